// Shared helpers for the gfx950 kernels behind include/pv_yield_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include <algorithm>
#include "../../include/pv_yield_hip.h"

// gfx950 only: the kernels size their LDS images for 160 KB per CU (linear_f32_skinny.hip alone declares 84 KB static), use
// gfx950's matrix and transposed-read instructions and 64-wide waves; another --offload-arch must fail here, not at a launch
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "predict_pv_yield_amd/csrc is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace pv {

// thread-local description of the last failure (pv_last_error()).
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PV_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return PV_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// opt-in per-stage device timing (stage_timer.hip): everything enqueued on `st` up to the next mark belongs to `label`
// (a string literal); nullptr closes the current stage.  A no-op unless pv_stage_timing_begin() armed it.
void stage_mark(const char* label, hipStream_t st);

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kNumCU = 256;    // MI355X

// grid for a grid-stride streaming kernel: enough blocks to fill 256 CUs x 8, no more.
inline unsigned stream_grid(size_t work_items, unsigned block) {
  size_t blocks = (work_items + block - 1) / block;
  size_t cap = (size_t)kNumCU * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// ---- bf16 helpers (storage type uint16_t at the ABI, __bf16 in kernels) ----
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  // plain cast: v_cvt_pk_bf16_f32 (RNE, NaN preserved) on gfx950
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
// two floats -> packed bf16 pair (a in the low half): ONE v_cvt_pk_bf16_f32 (RNE), where converting the halves one by one
// and or-ing them costs a convert, a shift and an or per value
typedef __attribute__((ext_vector_type(2))) float pv_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 pv_bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16_pair(float a, float b) {
  const pv_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pv_bf16x2_t));
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t u) {
  return __builtin_bit_cast(float, (uint32_t)u << 16);
}

// ReLU-derivative bits of a packed bf16 pair.  As 16-bit integers a bf16 is > 0 exactly when the signed value is
// (sign clear, not +0): clamp to [0, 1] with two packed ops (v_pk_max_i16, v_pk_min_u16).
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
__device__ __forceinline__ uint32_t relu_pair01(uint32_t g) {   // bit 0 = (low half > 0), bit 16 = (high half > 0)
  const s16x2_t pos = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, g), (s16x2_t){0, 0});
  const u16x2_t one = __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, pos), (u16x2_t){1, 1});
  return __builtin_bit_cast(uint32_t, one);
}
__device__ __forceinline__ uint32_t relu_bits_of_pair(uint32_t g) {   // bit 0 = (low half > 0), bit 1 = (high half > 0)
  const uint32_t t = relu_pair01(g);
  return (t | (t >> 15)) & 3u;
}
// the 8 relu bits (bit i = element i > 0) of 8 packed bf16
__device__ __forceinline__ uint32_t relu_bits_of_8(const uint32_t (&w)[4]) {
  const uint32_t t = relu_pair01(w[0]) | (relu_pair01(w[1]) << 2) | (relu_pair01(w[2]) << 4) | (relu_pair01(w[3]) << 6);
  return (t | (t >> 15)) & 0xffu;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// the mask byte of 8 consecutive channels held as four bf16 pairs: bit c = channel c > 0
__device__ __forceinline__ uint32_t relu_byte_of_pairs(const u32x4& v) {
  uint32_t u = 0;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    uint32_t m;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(m) : "v"(v[j]));
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "s"(0x00010001u));
    u = (u << 2) | m;      // halves: bits 2 j (low) and 16 + 2 j (high)
  }
  return u | (u >> 15);      // bit 2 j + 1 <- bit 16 + 2 j (the byte store keeps bits 0..7)
}


}  // namespace pv

// ---- diagnostic build only (make diag: -DPV_DIAG_STAMPS -> lib/libpvyield_diag.so, tools/diag_stamps.py) --------------------
// s_memtime stamps around the phases of a kernel's main loop; the per-phase sums leave through a buffer of their own that no
// kernel reads.  The product library is built without the macro: no stamp executes there.
#ifdef PV_DIAG_STAMPS
#define PV_STAMP(var)                                                                       \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#define PV_DIAG_SLOTS 8
#define PV_DIAG_WAVES (1 << 14)
#else
#define PV_STAMP(var) do { } while (0)
#endif

#define PV_REQUIRE(cond, code, ...) \
  do { if (!(cond)) return pv::fail(code, __VA_ARGS__); } while (0)
