// xcl2.hpp -- what callers on the SpMV / SpMSpV hot path take from the reference's
// xrt/includes/xcl2/xcl2.hpp, for the MI355X build:
//
//   * the global `aligned_allocator<T>` (xcl2.hpp:61-76) -- defined in graphlily/global.h, where the
//     module headers need it too; benchmark/bench_spmv.cpp:9,16-17,57-64,82 names it through this header;
//   * the print-and-exit error convention `OCL_CHECK(error, call)` (xcl2.hpp:40-46), with the status of this
//     backend (GL_OK == 0 plays CL_SUCCESS).
//
// Everything else in the reference header -- OpenCL device discovery, the xclbin reader, the Xilinx stream
// extension -- has no counterpart: there is no OpenCL runtime and no bitstream behind this backend.
#ifndef GRAPHLILY_HIP_XCL2_HPP_
#define GRAPHLILY_HIP_XCL2_HPP_

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "graphlily/global.h"

#ifndef CL_SUCCESS
#define CL_SUCCESS 0
#endif

// OCL_CHECK doesn't work if call has a templatized function call (same restriction as the reference's macro)
#define OCL_CHECK(error, call)                                                             \
    call;                                                                                  \
    if (error != CL_SUCCESS) {                                                             \
        printf("%s:%d Error calling " #call ", error code is: %d (%s)\n", __FILE__, __LINE__, \
               (int)(error), gl_last_error());                                             \
        exit(EXIT_FAILURE);                                                                \
    }

#endif  // GRAPHLILY_HIP_XCL2_HPP_
