// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "leco_hip.h"

namespace leco {
// thread-local message returned by leco_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);
// after a launch: translate hipGetLastError() into the ABI's return convention
int check_launch(const char* what);
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// gemm.hip: sums split-K partial slabs ws[splits][M][N] and applies the epilogue of `a`
void splitk_finish_launch(const leco_gemm_args& a, const float* ws, int splits, hipStream_t s);
// conv_patch.hip: patch-staged 3x3 / stride-1 convolution.  0 = launched (or described into `describe`), 1 = not
// applicable (caller falls back to the implicit-GEMM kernel), < 0 = error.
int conv_patch_try(const leco_gemm_args& a, int variant, int split_k, float* ws, hipStream_t s, char* describe,
                   int describe_len);
// conv_patch.hip: the patch variant (tile id 7..10) and split_k its cost model picks for `a`; 0 = not applicable
int conv_patch_choose(const leco_gemm_args& a, int64_t ws_bytes, int split_in, int* split);
}  // namespace leco
