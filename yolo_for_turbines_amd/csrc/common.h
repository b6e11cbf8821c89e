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

}  // namespace yolo
