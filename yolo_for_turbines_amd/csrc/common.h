// Shared host-side helpers for libyolo_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/yolo_mi355x.h"

namespace yolo {

// thread-local message returned by yolo_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(YOLO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return YOLO_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// packed-weight geometry (see yolo_packed_weight_elems in the header)
inline int cin_pad_of(int cin) { return round_up(cin, 4); }
inline int kpad_of(int cin, int ks) { return round_up(ks * ks * cin_pad_of(cin), 32); }
inline int coutpad_of(int cout) { return round_up(cout, 128); }


// conv_f32_v2.hip ("patch + fragment stream" kernel, stride 1, cin % 32 == 0)
bool v2_eligible(const yolo_conv_desc* d);
size_t v2_frag_elems(int cout, int cin, int ks);
int v2_blocks(const yolo_conv_desc* d, int bn);          // grid size the patch kernel would launch
int v2_pack(const float* w_oihw, float* wf, int cout, int cin, int ks, hipStream_t s);
int conv_v2_launch(const yolo_conv_desc* d, const void* x, const float* wf, const float* scale, const float* shift,
                   const void* residual, void* y, int32_t* nan_flag, int bn, hipStream_t s);
// conv_h16.hip (bf16 / fp16 patch kernel)
size_t h16_frag_elems(int cout, int cin, int ks);
int h16_pack(const float* w_oihw, void* wf, int cout, int cin, int ks, int dtype, hipStream_t s);
int conv_h16_launch(const yolo_conv_desc* d, const void* x, const void* wf, const float* scale, const float* shift,
                    const void* residual, void* y, int32_t* nan_flag, hipStream_t s);
// offset (elements) of the fragment-order copy inside a packed weight buffer
inline size_t v0_packed_elems(int cout, int cin, int ks) { return (size_t)coutpad_of(cout) * kpad_of(cin, ks); }

}  // namespace yolo
