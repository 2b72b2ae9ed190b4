// dn_cpx.hpp -- complex arithmetic on packed fp32 for gfx950.
//
// A complex number is one even-aligned VGPR pair (re, im).  CDNA's packed-fp32 VALU ops
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) take per-operand half-select (op_sel, op_sel_hi)
// and negate (neg_lo, neg_hi) modifiers, so multiplication by +-i, conjugation and the cross terms
// of a complex product are free operand modifiers instead of v_mov shuffles: a butterfly
// add/sub with a rotated operand is ONE instruction, a complex multiply is TWO.  hipcc does not
// find these forms from scalar source (it emitted ~25 % v_mov in the Griffin-Lim loop), hence the
// inline asm; all of it is plain VALU (no manual wait states needed).
//
// This header is included as <dn_cpx.hpp> (the build adds -I.), so a test build for another host can
// put an arithmetic-only header of the same name ahead of it on the include path.
#pragma once
#include <hip/hip_runtime.h>

typedef float v2f __attribute__((ext_vector_type(2)));
// operand fragment of v_mfma_f32_16x16x32_bf16: 8 bf16 (4 VGPRs); float -> bf16 is round-to-nearest-even (v_cvt_pk_bf16_f32)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ __bf16 f2bf(float x) { return (__bf16)x; }

namespace dn {

__device__ __forceinline__ v2f mk2(float re, float im) { return v2f{re, im}; }
__device__ __forceinline__ v2f cadd(v2f a, v2f b) { return a + b; }
__device__ __forceinline__ v2f csub(v2f a, v2f b) { return a - b; }
__device__ __forceinline__ v2f cscale(v2f a, float s) { return a * s; }

// elementwise product (re*re, im*im) that is ROUNDED: never contracted into a following add (window products that the three-wave
// Griffin-Lim rounds by storing them to LDS and the one-wave form keeps in registers: both must give the same bits)
__device__ __forceinline__ v2f cmul_elem(v2f a, v2f b) {
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (-i) b  =  (a.re + b.im, a.im - b.re)
__device__ __forceinline__ v2f cadd_mi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + (+i) b  =  (a.re - b.im, a.im + b.re)
__device__ __forceinline__ v2f cadd_pi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + rot(b), a - rot(b) with rot = multiplication by -i (forward transform) or +i (inverse)
template <bool INV> __device__ __forceinline__ v2f cadd_rot(v2f a, v2f b) { return INV ? cadd_pi(a, b) : cadd_mi(a, b); }
template <bool INV> __device__ __forceinline__ v2f csub_rot(v2f a, v2f b) { return INV ? cadd_mi(a, b) : cadd_pi(a, b); }

// a + conj(b), a - conj(b)
__device__ __forceinline__ v2f cadd_conj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f csub_conj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// a * b
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
// a * conj(b)
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(d) : "v"(a), "v"(b));
    return d;
}

// 0.5 a + (-i) b   and   0.5 a + (+i) b
__device__ __forceinline__ v2f chalf_add_mi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, 0.5, %2 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f chalf_add_pi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, 0.5, %2 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// conj(0.5 a + i b)  =  (0.5 a.re - b.im, -0.5 a.im - b.re)
__device__ __forceinline__ v2f cconj_half_add_pi(v2f a, v2f b) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, 0.5, %2 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// conj(0.5 a) + i conj(b)  =  (0.5 a.re + b.im, -0.5 a.im + b.re)
__device__ __forceinline__ v2f chalf_conj_add_iconj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, 0.5, %2 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

}  // namespace dn
