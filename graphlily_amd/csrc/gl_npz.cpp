// scipy.sparse.save_npz reader: zip central directory walk + zlib raw inflate +
// .npy header parse.  Replaces the un-vendored cnpy dependency behind
// load_csr_matrix_from_float_npz (io/data_loader.h:51-70).
//
// The reference decodes `shape` as the low 32-bit words of an int64 pair,
// `indices`/`indptr` as 32-bit words and `data` as float32 (data_loader.h:55-68).
// This reader produces the same values for such files and additionally accepts
// int64 index arrays (scipy switches to them for very large matrices) and
// float64 data by narrowing.
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <algorithm>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "gl_common.h"

namespace {

struct Npy {
    std::string descr;
    std::vector<uint64_t> shape;
    std::vector<uint8_t> raw;  // payload only (header stripped)
    size_t count() const {
        size_t n = 1;
        for (uint64_t d : shape) n *= (size_t)d;
        return n;
    }
    size_t itemsize() const { return descr.size() >= 3 ? (size_t)atoi(descr.c_str() + 2) : 0; }
};

template <typename T>
static T rd(const uint8_t *p) {
    T v;
    memcpy(&v, p, sizeof(T));
    return v;
}

static bool read_file(const char *path, std::vector<uint8_t> &buf, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        err = std::string("cannot open ") + path;
        return false;
    }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 22) {
        fclose(f);
        err = "file too small to be a zip archive";
        return false;
    }
    buf.resize((size_t)sz);
    size_t got = fread(buf.data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) {
        err = "short read";
        return false;
    }
    return true;
}

static bool inflate_raw(const uint8_t *src, size_t src_len, std::vector<uint8_t> &dst, size_t dst_len, std::string &err) {
    dst.resize(dst_len);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) {
        err = "inflateInit2 failed";
        return false;
    }
    size_t in_off = 0, out_off = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        size_t in_chunk = std::min<size_t>(src_len - in_off, 1u << 30);
        size_t out_chunk = std::min<size_t>(dst_len - out_off, 1u << 30);
        zs.next_in = const_cast<Bytef *>(src + in_off);
        zs.avail_in = (uInt)in_chunk;
        zs.next_out = dst.data() + out_off;
        zs.avail_out = (uInt)out_chunk;
        rc = inflate(&zs, Z_NO_FLUSH);
        in_off += in_chunk - zs.avail_in;
        out_off += out_chunk - zs.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END) break;
        if (rc == Z_OK && in_chunk == 0 && out_chunk == 0) break;
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || out_off != dst_len) {
        err = "inflate failed (corrupt member)";
        return false;
    }
    return true;
}

static bool parse_npy(std::vector<uint8_t> &blob, Npy &out, std::string &err) {
    if (blob.size() < 10 || memcmp(blob.data(), "\x93NUMPY", 6) != 0) {
        err = "bad .npy magic";
        return false;
    }
    uint8_t major = blob[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = rd<uint16_t>(&blob[8]);
        hoff = 10;
    } else {
        if (blob.size() < 12) { err = "truncated .npy"; return false; }
        hlen = rd<uint32_t>(&blob[8]);
        hoff = 12;
    }
    if (hoff + hlen > blob.size()) {
        err = "truncated .npy header";
        return false;
    }
    std::string h((const char *)&blob[hoff], hlen);
    size_t p = h.find("'descr'");
    if (p == std::string::npos) { err = "no descr"; return false; }
    p = h.find('\'', p + 7);
    size_t q = h.find('\'', p + 1);
    if (p == std::string::npos || q == std::string::npos) { err = "bad descr"; return false; }
    out.descr = h.substr(p + 1, q - p - 1);
    if (h.find("'fortran_order': True") != std::string::npos) { err = "fortran order not supported"; return false; }
    p = h.find("'shape'");
    if (p == std::string::npos) { err = "no shape"; return false; }
    p = h.find('(', p);
    q = h.find(')', p);
    if (p == std::string::npos || q == std::string::npos) { err = "bad shape"; return false; }
    out.shape.clear();
    uint64_t cur = 0;
    bool have = false;
    for (size_t i = p + 1; i < q; i++) {
        char c = h[i];
        if (c >= '0' && c <= '9') { cur = cur * 10 + (uint64_t)(c - '0'); have = true; }
        else if (have) { out.shape.push_back(cur); cur = 0; have = false; }
    }
    if (have) out.shape.push_back(cur);
    out.raw.assign(blob.begin() + (long)(hoff + hlen), blob.end());
    if (out.descr.size() >= 2 && out.descr[0] == '>') { err = "big-endian arrays not supported"; return false; }
    if (out.descr != "|S3" && out.raw.size() < out.count() * out.itemsize()) { err = "truncated .npy payload"; return false; }
    return true;
}

static bool load_npz(const char *path, std::map<std::string, Npy> &arrays, std::string &err) {
    std::vector<uint8_t> f;
    if (!read_file(path, f, err)) return false;
    // end of central directory
    size_t eocd = std::string::npos;
    if (f.size() < 22) { err = "file too short to be a zip archive"; return false; }
    for (size_t i = f.size() - 22;; i--) {
        if (rd<uint32_t>(&f[i]) == 0x06054b50u) { eocd = i; break; }
        if (i == 0 || f.size() - i > 65557) break;
    }
    if (eocd == std::string::npos) { err = "zip end-of-central-directory not found"; return false; }
    uint64_t nent = rd<uint16_t>(&f[eocd + 10]);
    uint64_t cd_off = rd<uint32_t>(&f[eocd + 16]);
    if ((nent == 0xffff || cd_off == 0xffffffffu) && eocd >= 20 && rd<uint32_t>(&f[eocd - 20]) == 0x07064b50u) {
        uint64_t z64 = rd<uint64_t>(&f[eocd - 20 + 8]);
        if (z64 <= f.size() && z64 + 56 <= f.size() && rd<uint32_t>(&f[z64]) == 0x06064b50u) {
            nent = rd<uint64_t>(&f[z64 + 32]);
            cd_off = rd<uint64_t>(&f[z64 + 48]);
        }
    }
    // every length below comes from the file: nothing is read or sized before it is checked against the file's end
    if (cd_off > f.size()) { err = "zip central directory offset beyond the file"; return false; }
    size_t p = (size_t)cd_off;
    for (uint64_t e = 0; e < nent; e++) {
        if (p > f.size() || f.size() - p < 46 || rd<uint32_t>(&f[p]) != 0x02014b50u) { err = "bad zip central directory"; return false; }
        uint16_t method = rd<uint16_t>(&f[p + 10]);
        uint64_t csize = rd<uint32_t>(&f[p + 20]), usize = rd<uint32_t>(&f[p + 24]);
        uint16_t nlen = rd<uint16_t>(&f[p + 28]), xlen = rd<uint16_t>(&f[p + 30]), clen = rd<uint16_t>(&f[p + 32]);
        uint64_t lho = rd<uint32_t>(&f[p + 42]);
        if (f.size() - p - 46 < (size_t)nlen + xlen + clen) { err = "truncated zip central directory entry"; return false; }
        std::string name((const char *)&f[p + 46], nlen);
        // zip64 extra field
        size_t x = p + 46 + nlen;
        const size_t xend = x + xlen;   // <= f.size(), checked above
        while (x + 4 <= xend) {
            uint16_t id = rd<uint16_t>(&f[x]), sz = rd<uint16_t>(&f[x + 2]);
            const size_t fend = std::min(x + 4 + (size_t)sz, xend);   // a field may not run past the extra block
            if (id == 0x0001) {
                size_t y = x + 4;
                if (usize == 0xffffffffu && y + 8 <= fend) { usize = rd<uint64_t>(&f[y]); y += 8; }
                if (csize == 0xffffffffu && y + 8 <= fend) { csize = rd<uint64_t>(&f[y]); y += 8; }
                if (lho == 0xffffffffu && y + 8 <= fend) { lho = rd<uint64_t>(&f[y]); y += 8; }
            }
            x += 4 + (size_t)sz;
        }
        p = xend + clen;
        if (lho > f.size() || f.size() - lho < 30 || rd<uint32_t>(&f[lho]) != 0x04034b50u) { err = "bad zip local header"; return false; }
        const uint64_t data64 = lho + 30 + rd<uint16_t>(&f[lho + 26]) + rd<uint16_t>(&f[lho + 28]);
        if (data64 > f.size() || csize > f.size() - data64) { err = "zip member exceeds file"; return false; }
        const size_t data = (size_t)data64;
        // deflate expands by at most ~1032x: a larger claim is not an array this file can hold
        if (usize > (uint64_t)csize * 1032u + 65536u) { err = "zip member claims an impossible uncompressed size"; return false; }
        std::vector<uint8_t> blob;
        if (method == 0) {
            blob.assign(f.begin() + (long)data, f.begin() + (long)(data + csize));
        } else if (method == 8) {
            if (!inflate_raw(&f[data], (size_t)csize, blob, (size_t)usize, err)) return false;
        } else {
            err = "unsupported zip compression method";
            return false;
        }
        if (name.size() > 4 && name.substr(name.size() - 4) == ".npy") name.resize(name.size() - 4);
        Npy a;
        if (!parse_npy(blob, a, err)) { err = name + ": " + err; return false; }
        arrays[name] = std::move(a);
    }
    return true;
}

static bool to_u32(const Npy &a, size_t n, uint32_t *dst, std::string &err) {
    if (a.count() < n) { err = "array shorter than expected"; return false; }
    if (a.descr == "<i4" || a.descr == "<u4") {
        memcpy(dst, a.raw.data(), n * 4);
    } else if (a.descr == "<i8" || a.descr == "<u8") {
        for (size_t i = 0; i < n; i++) dst[i] = (uint32_t)rd<uint64_t>(&a.raw[i * 8]);
    } else {
        err = "unsupported index dtype " + a.descr;
        return false;
    }
    return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// io::csr2csc (io/data_loader.h:108-144) as a parallel counting sort: threads own contiguous row
// ranges, so inside a column entries keep ascending row order exactly like the reference's loop.
int gl::host_csr2csc(uint32_t num_rows, uint32_t num_cols, const uint32_t *indptr, const uint32_t *indices,
                               const float *data, uint32_t *csc_indptr, uint32_t *csc_indices, float *csc_data) {
    GL_ARG(indptr != nullptr && csc_indptr != nullptr);
    const uint64_t nnz = indptr[num_rows];
    GL_ARG(nnz == 0 || (indices != nullptr && data != nullptr && csc_indices != nullptr && csc_data != nullptr));
    for (uint64_t i = 0; i < nnz; i++)
        if (indices[i] >= num_cols)
            return gl::set_error(GL_ERR_INVALID_ARG, "gl_csr2csc: column index %u out of range (num_cols %u)", indices[i], num_cols);
    int T = 1;
#ifdef _OPENMP
    T = std::min(omp_get_max_threads(), 16);
#endif
    if (nnz < (1u << 20)) T = 1;
    // row range of thread t: balanced by nnz
    std::vector<uint32_t> rb(T + 1, num_rows);
    rb[0] = 0;
    for (int t = 1; t < T; t++) {
        const uint64_t want = nnz * t / T;
        rb[t] = (uint32_t)(std::lower_bound(indptr, indptr + num_rows + 1, (uint32_t)want) - indptr);
        if (rb[t] < rb[t - 1]) rb[t] = rb[t - 1];
    }
    std::vector<std::vector<uint32_t>> cnt(T, std::vector<uint32_t>(num_cols, 0));
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; t++)
        for (uint64_t i = indptr[rb[t]]; i < indptr[rb[t + 1]]; i++) cnt[t][indices[i]]++;
    // column-major prefix over (column, thread): thread t's entries of column c start at off[t][c]
    uint64_t run = 0;
    for (uint32_t c = 0; c < num_cols; c++) {
        csc_indptr[c] = (uint32_t)run;
        for (int t = 0; t < T; t++) {
            const uint32_t k = cnt[t][c];
            cnt[t][c] = (uint32_t)run;
            run += k;
        }
    }
    csc_indptr[num_cols] = (uint32_t)run;
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; t++)
        for (uint32_t r = rb[t]; r < rb[t + 1]; r++)
            for (uint64_t i = indptr[r]; i < indptr[r + 1]; i++) {
                const uint32_t dst = cnt[t][indices[i]]++;
                csc_indices[dst] = r;
                csc_data[dst] = data[i];
            }
    return GL_OK;
}

struct gl_npz_csr_s {
    std::map<std::string, Npy> arrays;
    uint32_t num_rows = 0, num_cols = 0;
    uint64_t nnz = 0;
};

extern "C" {

int gl_npz_csr_open(const char *path, gl_npz_csr *handle, uint32_t *num_rows, uint32_t *num_cols, uint64_t *nnz) {
    GL_ARG(path != nullptr && handle != nullptr);
    gl_npz_csr h = new gl_npz_csr_s();
    std::string err;
    if (!load_npz(path, h->arrays, err)) {
        delete h;
        return gl::set_error(GL_ERR_IO, "gl_npz_csr_open(%s): %s", path, err.c_str());
    }
    for (const char *k : {"shape", "data", "indices", "indptr"}) {
        if (!h->arrays.count(k)) {
            delete h;
            return gl::set_error(GL_ERR_IO, "gl_npz_csr_open(%s): member '%s' missing", path, k);
        }
    }
    if (h->arrays.count("format")) {
        const Npy &fm = h->arrays["format"];
        if (fm.raw.size() >= 3 && memcmp(fm.raw.data(), "csr", 3) != 0) {
            delete h;
            return gl::set_error(GL_ERR_IO, "gl_npz_csr_open(%s): sparse format is not csr", path);
        }
    }
    const Npy &sh = h->arrays["shape"];
    uint32_t dims[2];
    if (sh.count() != 2 || !to_u32(sh, 2, dims, err)) {
        delete h;
        return gl::set_error(GL_ERR_IO, "gl_npz_csr_open(%s): bad shape array", path);
    }
    h->num_rows = dims[0];
    h->num_cols = dims[1];
    const Npy &da = h->arrays["data"];
    if (da.descr != "<f4" && da.descr != "<f8") {
        std::string d = da.descr;
        delete h;
        return gl::set_error(GL_ERR_IO, "gl_npz_csr_open(%s): data dtype %s is not float", path, d.c_str());
    }
    h->nnz = da.count();
    if (h->arrays["indices"].count() < h->nnz || h->arrays["indptr"].count() < (size_t)h->num_rows + 1) {
        delete h;
        return gl::set_error(GL_ERR_IO, "gl_npz_csr_open(%s): indices/indptr shorter than data/shape imply", path);
    }
    *handle = h;
    if (num_rows) *num_rows = h->num_rows;
    if (num_cols) *num_cols = h->num_cols;
    if (nnz) *nnz = h->nnz;
    return GL_OK;
}

int gl_npz_csr_read(gl_npz_csr h, float *data, uint32_t *indices, uint32_t *indptr) {
    GL_ARG(h != nullptr);
    std::string err;
    bool ok = true;
    if (data) {
        const Npy &da = h->arrays["data"];
        if (da.descr == "<f4") {
            memcpy(data, da.raw.data(), h->nnz * 4);
        } else {
            for (uint64_t i = 0; i < h->nnz; i++) data[i] = (float)rd<double>(&da.raw[i * 8]);
        }
    }
    if (ok && indices) ok = to_u32(h->arrays["indices"], h->nnz, indices, err);
    if (ok && indptr) ok = to_u32(h->arrays["indptr"], (size_t)h->num_rows + 1, indptr, err);
    delete h;
    if (!ok) return gl::set_error(GL_ERR_IO, "gl_npz_csr_read: %s", err.c_str());
    return GL_OK;
}

int gl_npz_csr_close(gl_npz_csr h) {
    delete h;
    return GL_OK;
}

}  // extern "C"
