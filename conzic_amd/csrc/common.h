// Shared device/host helpers for the gfx950 kernels of the polishing engine.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdio.h>

namespace czc {

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__host__ __device__ __forceinline__ float bf2f(bf16_t h) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)h) << 16;
  return c.f;
}
// round-to-nearest-even, NaN preserved
__host__ __device__ __forceinline__ bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair with the hardware converter (round-to-nearest-even like f2bf; one instruction
// instead of ~8 plus a divergent NaN branch -- the software form was 2/3 of the attention kernels' VALU work)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// Second 2-byte operand type: IEEE fp16 (engine precision CZC_PREC_FP16).  Same storage size, MFMA rate, LDS images
// and DMA patterns as bf16 -- every bf16 kernel is instantiated for it by swapping the MFMA opcode and the converter
// (Half<> below) -- with 11 significand bits instead of 8: the CLIP cosine error drops ~8x, which keeps the fused
// score inside the 1e-3 budget at the published checkpoints' logit scale (x100) where bf16 is out by 2-3x.  All fp16
// operands of the towers are bounded (LayerNorm outputs, attention context, quick-GELU outputs; residual stream,
// accumulators, softmax and LayerNorm statistics stay fp32); an overflow would surface as a non-finite cosine, which the
// combine kernel reports.
struct f16_t { unsigned short v; };
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;

// fp32 -> fp16 conversions must start from ONE fp32 value.  HIP compiles with -ffp-contract=fast: when v is the result of
// a multiply or add that is still visible (o * inv, acc + bias, (x - mean) * rstd * g + b), the compiler may convert with a
// single-rounding v_fma_mix*_f16 from the exact product in one place and with v_cvt_pk_f16_f32 from the fp32-rounded
// value in another; at near-ties (about 1 value in 8000) the two differ by one fp16 ulp.  That broke the split
// representation (the hi that feeds `v - hi` against the stored hi: hi + lo off by 2^-11 relative) and makes two kernels
// that state the same arithmetic disagree in the last place.  pin() makes the fp32 value opaque first.
__device__ __forceinline__ float pin(float v) {
  asm("" : "+v"(v));
  return v;
}

// (the products are made opaque: __fmul_rn does not stop hipcc from contracting a * a + b * b into a fused multiply-add,
// and it fuses differently from kernel to kernel)
__device__ __forceinline__ float ln_sq4(float a, float b, float c, float d) {
  return __fadd_rn(__fadd_rn(pin(a * a), pin(b * b)), __fadd_rn(pin(c * c), pin(d * d)));
}

__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const h2 v = {(_Float16)pin(lo), (_Float16)pin(hi)};  // round-to-nearest-even of the fp32 values (v_cvt_pk_f16_f32)
  return __builtin_bit_cast(uint32_t, v);
}

// HT = bf16_t | f16_t: what a kernel written on raw 16-bit storage needs to know about its element type
template <typename HT> struct Half;
template <> struct Half<bf16_t> {
  template <typename V>
  __device__ static __forceinline__ f32x16_t mfma(const V& a, const V& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2_bf16(lo, hi); }
  __host__ __device__ static __forceinline__ float to_f32(unsigned short r) { return bf2f(r); }
  __host__ __device__ static __forceinline__ unsigned short from_f32(float f) { return f2bf(f); }
};
template <> struct Half<f16_t> {
  template <typename V>
  __device__ static __forceinline__ f32x16_t mfma(const V& a, const V& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2_f16(lo, hi); }
  __device__ static __forceinline__ float to_f32(unsigned short r) { return (float)__builtin_bit_cast(_Float16, r); }
  __device__ static __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)pin(f)); }
};

// activation storage type per engine precision.  All accessors take (base pointer, ELEMENT index):
// for the linear types that is base[idx]; split_t stores an fp32-sized element as two fp16 planes
// in groups of 8 elements -- 16 bytes of hi parts, then 16 bytes of lo parts (v ~ hi + lo, 22
// mantissa bits) -- so that an MFMA fragment (8 consecutive k) is one 16-byte load per plane.
// Row lengths must be multiples of 8 elements.
struct split_t { unsigned int raw; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

template <typename T> struct Act;
template <> struct Act<bf16_t> {
  __device__ static __forceinline__ float ld(const bf16_t* b, long i) { return bf2f(b[i]); }
  __device__ static __forceinline__ void st(bf16_t* b, long i, float v) { b[i] = f2bf(v); }
  __device__ static __forceinline__ void st4(bf16_t* b, long i, float x, float y, float z, float w) {
    uint2 o;
    o.x = pack2_bf16(x, y);
    o.y = pack2_bf16(z, w);
    *(uint2*)(b + i) = o;
  }
};
template <> struct Act<f16_t> {
  __device__ static __forceinline__ float ld(const f16_t* b, long i) { return Half<f16_t>::to_f32(b[i].v); }
  __device__ static __forceinline__ void st(f16_t* b, long i, float v) { b[i].v = Half<f16_t>::from_f32(v); }
  __device__ static __forceinline__ void st4(f16_t* b, long i, float x, float y, float z, float w) {
    uint2 o;
    o.x = pack2_f16(x, y);
    o.y = pack2_f16(z, w);
    *(uint2*)(b + i) = o;
  }
};
template <> struct Act<float> {
  __device__ static __forceinline__ float ld(const float* b, long i) { return b[i]; }
  __device__ static __forceinline__ void st(float* b, long i, float v) { b[i] = v; }
  __device__ static __forceinline__ void st4(float* b, long i, float x, float y, float z, float w) {
    *(float4*)(b + i) = make_float4(x, y, z, w);
  }
};
template <> struct Act<split_t> {
  __device__ static __forceinline__ long off(long i) { return (i >> 3) * 32 + (i & 7) * 2; }
  __device__ static __forceinline__ float ld(const split_t* b, long i) {
    const unsigned char* p = (const unsigned char*)b + off(i);
    return (float)*(const _Float16*)p + (float)*(const _Float16*)(p + 16);
  }
  __device__ static __forceinline__ void st(split_t* b, long i, float v) {
    unsigned char* p = (unsigned char*)b + off(i);
    v = pin(v);
    const _Float16 hi = (_Float16)v;
    *(_Float16*)p = hi;
    *(_Float16*)(p + 16) = (_Float16)(v - (float)hi);
  }
  __device__ static __forceinline__ void st4(split_t* b, long i, float x, float y, float z, float w) {  // i % 4 == 0
    unsigned char* p = (unsigned char*)b + off(i);
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    x = pin(x); y = pin(y); z = pin(z); w = pin(w);
    h4 hi = {(_Float16)x, (_Float16)y, (_Float16)z, (_Float16)w};
    h4 lo = {(_Float16)(x - (float)hi[0]), (_Float16)(y - (float)hi[1]), (_Float16)(z - (float)hi[2]),
             (_Float16)(w - (float)hi[3])};
    *(h4*)p = hi;
    *(h4*)(p + 16) = lo;
  }
};

// LayerNorm partials of four stored values (GemmArgs::row_part): fixed association, no fused multiply-adds (HIP compiles with
// -ffp-contract=fast, and every producer kernel must arrive at the same bits)
__device__ __forceinline__ float ln_sum4(float a, float b, float c, float d) { return __fadd_rn(__fadd_rn(a, b), __fadd_rn(c, d)); }
__device__ __forceinline__ float f16lo(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float f16hi(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

enum { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU_ERF = 2 };
// 3: split_t storage, three fp16 MFMA passes (~fp32 accuracy); 4: f16_t storage, one fp16 MFMA pass (the bf16 kernels on fp16)
enum { PREC_BF16 = 0, PREC_F32 = 1, PREC_F16X3 = 3, PREC_F16 = 4 };
__host__ __device__ inline bool prec_is_half(int p) { return p == PREC_BF16 || p == PREC_F16; }
__host__ __device__ inline size_t prec_bytes(int p) { return prec_is_half(p) ? 2 : 4; }

#define CZC_HIP_CHECK(expr)                                                                      \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      snprintf(czc::g_err, sizeof(czc::g_err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr,      \
               hipGetErrorString(_e));                                                           \
      return 2;                                                                                  \
    }                                                                                            \
  } while (0)

extern thread_local char g_err[512];  // per host thread: two engines of an EngineGroup launch from two threads

// One-time launch set-up of a kernel family (CU count, dynamic-LDS attributes), run by whichever host thread gets
// there first behind a function-local static (C++11: initialised exactly once, other threads wait).
struct LaunchInit { int n_cu = 0; int rc = 0; };
template <typename F>
inline LaunchInit launch_init(F attrs) {
  LaunchInit li;
  li.rc = [&]() -> int {
    int dev = 0;
    hipDeviceProp_t prop;
    CZC_HIP_CHECK(hipGetDevice(&dev));
    CZC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    li.n_cu = prop.multiProcessorCount;
    return attrs(li);
  }();
  return li;
}
// The same, once per DEVICE: hipFuncSetAttribute and the CU count belong to the device that is current when they are
// made, and czc_create takes a device id -- engines on two GPUs of one process each get their own set-up.  A failed
// set-up is not latched: the next launch on that device tries again.
struct PerDeviceInit {
  static constexpr int MAX_DEV = 32;
  std::mutex mu;
  std::atomic<bool> done[MAX_DEV];
  LaunchInit li[MAX_DEV];
  PerDeviceInit() { for (auto& d : done) d.store(false); }
  template <typename F>
  LaunchInit get(F attrs) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) { LaunchInit bad; bad.rc = 2; return bad; }
    if (done[dev].load(std::memory_order_acquire)) return li[dev];
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev].load(std::memory_order_relaxed)) return li[dev];
    LaunchInit r = launch_init(attrs);
    if (r.rc == 0) { li[dev] = r; done[dev].store(true, std::memory_order_release); }
    return r;
  }
};
inline int launch_init_failed(const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: one-time launch set-up failed (hipFuncSetAttribute / device query)", what);
  return 2;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace czc
