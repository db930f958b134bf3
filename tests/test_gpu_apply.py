"""GPU parity: the element-wise apply modules vs the CPU oracle, following tests/test_module_apply.cpp."""
import numpy as np
import pytest

from graphlily_amd import capi, module as M
from oracle import oracle as O

from helpers import rand01

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("length", [128, 8, 1000003])
def test_add_scalar_vector_dense(gpu, length):
    # TEST(AddScalarVectorDense, Basic) :54-75 (length 128, val 1, in = (rand()%10)/100)
    inp = (np.random.default_rng(0).integers(0, 10, size=length) / 100.0).astype(np.float32)
    mod = M.eWiseAddModule()
    mod.set_up_runtime()
    mod.send_in_host_to_device(inp)
    mod.allocate_out_buf(length)
    mod.run(length, 1.0)
    assert np.array_equal(mod.send_out_device_to_host(), O.ewise_add(inp, length, 1.0))


@pytest.mark.parametrize("mask_type", [M.kMaskWriteToOne, M.kMaskWriteToZero])
@pytest.mark.parametrize("length", [128, 100001])
def test_assign_vector_dense(gpu, mask_type, length):
    # TEST(AssignVectorDense, Basic) :78-103 (length 128, val 23, WriteToOne)
    mask, inout = rand01(length, 1), rand01(length, 2)
    mod = M.AssignVectorDenseModule()
    mod.set_up_runtime()
    mod.set_mask_type(mask_type)
    mod.send_mask_host_to_device(mask)
    mod.send_inout_host_to_device(inout)
    mod.run(length, 23.0)
    ref = inout.copy()
    O.assign_dense(mask_type, mask, ref, length, 23.0)
    assert np.array_equal(mod.send_inout_device_to_host(), ref)


def test_assign_vector_dense_nomask_is_fatal(gpu):
    mod = M.AssignVectorDenseModule()
    with pytest.raises(SystemExit):
        mod.set_mask_type(M.kNoMask)          # assign_vector_dense_module.h:88-95
    with pytest.raises(capi.GraphLilyError) as e:
        capi.assign_dense(capi.DeviceBuffer(32), capi.DeviceBuffer(32), 8, 1.0, capi.GL_NOMASK)
    assert e.value.code == capi.GL_ERR_INVALID_ARG


def _strided_mask(inout_size, sparsity, seed):
    length = int(np.floor(inout_size * (1 - sparsity)))
    inc = inout_size // length
    vals = np.random.default_rng(seed).integers(0, 10, size=length).astype(np.float32)
    return M.make_sparse_vec(np.arange(length, dtype=np.uint32) * inc, vals)


@pytest.mark.parametrize("inout_size", [8192, 500000])
def test_assign_vector_sparse_no_new_frontier(gpu, inout_size):
    # TEST(AssignVectorSparseNoNewFrontier, Basic) :106-143 (n 8192, 10% dense mask, val 3)
    mask = _strided_mask(inout_size, 0.9, 0)
    inout = np.random.default_rng(1).integers(0, 10, size=inout_size).astype(np.float32)
    mod = M.AssignVectorSparseModule(False)
    mod.set_up_runtime()
    mod.send_mask_host_to_device(mask)
    mod.send_inout_host_to_device(inout)
    mod.run(3.0)
    ref = inout.copy()
    O.assign_sparse(mask, ref, 3.0)
    assert np.array_equal(mod.send_inout_device_to_host(), ref)
    with pytest.raises(SystemExit):
        mod.run()                              # wrong mode exits (assign_vector_sparse_module.h:296-300)


@pytest.mark.parametrize("inout_size,inf", [(128, 255.0), (300000, 999999999.0)])
def test_assign_vector_sparse_new_frontier(gpu, inout_size, inf):
    # TEST(AssignVectorSparseNewFrontier, Basic) :146-206 (n 128, inout in {5, inf})
    mask = _strided_mask(inout_size, 0.9, 2)
    inout = np.where(np.random.default_rng(3).integers(0, 10, size=inout_size) > 5, 5.0, inf).astype(np.float32)
    mod = M.AssignVectorSparseModule(True)
    mod.set_up_runtime()
    mod.send_mask_host_to_device(mask)
    mod.send_inout_host_to_device(inout)
    mod.run()
    ref = inout.copy()
    ref_nf = O.assign_sparse_new_frontier(mask, ref)
    assert np.array_equal(mod.send_inout_device_to_host(), ref)
    nf = mod.send_new_frontier_device_to_host()
    n = int(nf["index"][0])
    assert n == int(ref_nf["index"][0]) and nf["val"][0] == 0.0
    # this build keeps mask order, so the list itself (not just its densification) matches
    assert np.array_equal(nf[:n + 1], ref_nf)
    with pytest.raises(SystemExit):
        mod.run(1.0)


def test_copy_buffer_bind_buffer(gpu):
    # TEST(CopyBufferBindBuffer, Basic) :209-261
    length = 128
    mask, inout = rand01(length, 4), np.zeros(length, np.float32)
    mod = M.AssignVectorDenseModule()
    mod.set_up_runtime()
    mod.set_mask_type(M.kMaskWriteToOne)
    mod.send_mask_host_to_device(mask)
    mod.send_inout_host_to_device(inout)
    mod.copy_buffer_device_to_device(mod.mask_buf, mod.inout_buf, 4 * length)
    assert np.array_equal(mod.send_inout_device_to_host(), mask)
    x_buf = capi.DeviceBuffer.from_host(np.zeros(length, np.float32))
    mod.send_mask_host_to_device(mask)
    mod.bind_inout_buf(x_buf)
    mod.run(length, 2.0)
    ref = np.zeros(length, np.float32)
    O.assign_dense(O.WRITETOONE, mask, ref, length, 2.0)
    assert np.array_equal(x_buf.read(np.float32), ref)


def test_sparse_to_dense(gpu):
    n = 100000
    sv = _strided_mask(n, 0.97, 5)
    cap = capi.DeviceBuffer(8 * (n + 1))
    cap.write(sv)
    dense = capi.DeviceBuffer(4 * n)
    capi.sparse_to_dense(cap, dense, n, 255.0, n)
    capi.sync()
    assert np.array_equal(dense.read(np.float32), O.convert_sparse_vec_to_dense_vec(sv, n, 255.0))


def test_pinned_readback(gpu):
    """gl_host_alloc: page-locked destination for device->host copies (DeviceBuffer.read(out=...))."""
    n = 300000
    src = np.arange(n, dtype=np.float32)
    buf = capi.DeviceBuffer.from_host(src)
    out = capi.pinned_empty(n, np.float32)
    got = buf.read(np.float32, n, out=out)
    assert np.array_equal(got, src) and np.shares_memory(got, out)
