// <ap_fixed.h> for callers of the GraphLily module API on the HIP backend.
//
// The reference's sources include Xilinx' <ap_fixed.h> for ONE type: graphlily::val_t = ap_ufixed<32, 8, AP_RND, AP_SAT>
// (graphlily/global.h:7, :63), and its tests and drivers include the header by name (tests/test_module_apply.cpp:12,
// tests/test_module_spmv_spmspv.cpp:11, tests/test_app.cpp:13).  This backend needs no HLS library: that one type is
// graphlily::ufixed_32_8 (include/graphlily/global.h -- a 32-bit word with 24 fraction bits, round-half-up conversion from
// double, saturation at 2^32 - 1; the kernels compute on the bits), and this header maps the Xilinx spelling onto it so that the
// reference's files compile UNMODIFIED.  Nothing else of ap_fixed.h exists here: any other instantiation is a compile error,
// on purpose.
#ifndef GRAPHLILY_HIP_AP_FIXED_H_
#define GRAPHLILY_HIP_AP_FIXED_H_

#include "graphlily/global.h"

enum ap_q_mode { AP_RND, AP_RND_ZERO, AP_RND_MIN_INF, AP_RND_INF, AP_RND_CONV, AP_TRN, AP_TRN_ZERO };
enum ap_o_mode { AP_SAT, AP_SAT_ZERO, AP_SAT_SYM, AP_WRAP, AP_WRAP_SM };

namespace graphlily_detail {
template <int W, int I, ap_q_mode Q, ap_o_mode O>
struct ap_ufixed_of;   // (only the reference's val_t is defined)
template <>
struct ap_ufixed_of<32, 8, AP_RND, AP_SAT> {
    typedef graphlily::ufixed_32_8 type;
};
}  // namespace graphlily_detail

template <int W, int I, ap_q_mode Q = AP_TRN, ap_o_mode O = AP_WRAP>
using ap_ufixed = typename graphlily_detail::ap_ufixed_of<W, I, Q, O>::type;

#endif  // GRAPHLILY_HIP_AP_FIXED_H_
