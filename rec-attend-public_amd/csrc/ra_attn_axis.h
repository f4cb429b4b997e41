// The separable Gaussian attention filter of modellib.get_gaussian_filter (modellib.py:581-612) evaluated on the
// fly from an attention record, shared by the decode kernels (ra_attn_direct.hip) and the training kernels
// (ra_attn_train.hip).
#pragma once
#include "ra_common.h"

namespace ra {
namespace attnd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kBandLog = 30.0f;
constexpr float kInvSqrt2Pi = 0.3989422804014327f;

struct Axis {  // one axis of one example's filter bank
  float ctr, step, inv_step, half, inv2var, norm, R;
  int L, F;
  __device__ inline float mu(int j) const { return ctr + step * ((float)j - half); }
  __device__ inline float w(float l, int j) const {  // modellib.py:610-611
    const float d = l - mu(j);
    return norm * __expf(-d * d * inv2var);
  }
  // pixel band [lo, hi) of tap j
  __device__ inline void band(int j, int &lo, int &hi) const {
    const float m = mu(j);
    float a = ceilf(m - R), c = floorf(m + R) + 1.0f;
    a = fminf(fmaxf(a, 0.0f), (float)L);
    c = fminf(fmaxf(c, 0.0f), (float)L);
    if (!(a == a) || !(c == c)) {
      a = 0.0f;
      c = (float)L;
    }
    lo = (int)a;
    hi = (int)c > lo ? (int)c : lo;
  }
  // tap range [jlo, jhi) whose band contains pixel l (widened by one tap each side: extra terms
  // are harmless, missing ones are not)
  __device__ inline void taps(int l, int &jlo, int &jhi) const {
    float a = ((float)l - R - ctr) * inv_step + half, c = ((float)l + R - ctr) * inv_step + half;
    if (!(a == a) || !(c == c) || !(step > 0.0f)) {
      jlo = 0;
      jhi = F;
      return;
    }
    a = fminf(fmaxf(floorf(a) - 1.0f, 0.0f), (float)F);
    c = fminf(fmaxf(ceilf(c) + 2.0f, 0.0f), (float)F);
    jlo = (int)a;
    jhi = (int)c;
  }
};

__device__ inline Axis make_axis(const float *rec, int axis, int L, int F) {
  // every workgroup evaluates this on its critical path: single-instruction reciprocal / square roots (1 ulp)
  Axis A;
  const float var = __expf(rec[4 + axis]);
  A.ctr = rec[0 + axis];
  A.step = (rec[2 + axis] + 1.0f) / (float)F;       // modellib.py:599
  A.inv_step = __builtin_amdgcn_rcpf(A.step);
  A.half = ((float)F - 1.0f) / 2.0f;
  A.inv2var = 0.5f * __builtin_amdgcn_rcpf(var);
  A.norm = kInvSqrt2Pi * __builtin_amdgcn_rsqf(var);  // 1/sqrt(var)/sqrt(2 pi)
  A.R = __builtin_amdgcn_sqrtf(2.0f * kBandLog * var);
  A.L = L;
  A.F = F;
  return A;
}

__device__ inline float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

}  // namespace attnd
}  // namespace ra
