// TEST INFRASTRUCTURE: arithmetic-only stand-in for audio-denoising_amd/csrc/dn_cpx.hpp (which uses
// gfx950 packed-math inline asm).  Same names, same meaning, plain C++; found first on the include
// path of the host-emulation build only.
#pragma once
#include <hip/hip_runtime.h>

typedef float v2f __attribute__((vector_size(8)));

// bf16 operand fragments for the emulated v_mfma_f32_16x16x32_bf16
struct dn_emu_bf16 {
    uint16_t bits;
    float to_float() const { uint32_t u = (uint32_t)bits << 16; float f; memcpy(&f, &u, 4); return f; }
};
inline dn_emu_bf16 f2bf(float x) {               // round to nearest even
    uint32_t u; memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return dn_emu_bf16{(uint16_t)(u >> 16)};
}
struct bf16x8 {
    dn_emu_bf16 v[8];
    dn_emu_bf16& operator[](int i) { return v[i]; }
    const dn_emu_bf16& operator[](int i) const { return v[i]; }
};

namespace dn {
inline v2f mk2(float re, float im) { return v2f{re, im}; }
inline v2f cadd(v2f a, v2f b) { return a + b; }
inline v2f csub(v2f a, v2f b) { return a - b; }
inline v2f cscale(v2f a, float s) { return v2f{a[0] * s, a[1] * s}; }
inline v2f cmul_elem(v2f a, v2f b) { volatile float r0 = a[0] * b[0], r1 = a[1] * b[1]; return v2f{r0, r1}; }
inline v2f cadd_mi(v2f a, v2f b) { return v2f{a[0] + b[1], a[1] - b[0]}; }
inline v2f cadd_pi(v2f a, v2f b) { return v2f{a[0] - b[1], a[1] + b[0]}; }
template <bool INV> inline v2f cadd_rot(v2f a, v2f b) { return INV ? cadd_pi(a, b) : cadd_mi(a, b); }
template <bool INV> inline v2f csub_rot(v2f a, v2f b) { return INV ? cadd_mi(a, b) : cadd_pi(a, b); }
inline v2f cadd_conj(v2f a, v2f b) { return v2f{a[0] + b[0], a[1] - b[1]}; }
inline v2f csub_conj(v2f a, v2f b) { return v2f{a[0] - b[0], a[1] + b[1]}; }
inline v2f cmul(v2f a, v2f b) { return v2f{std::fma(a[1], -b[1], a[0] * b[0]), std::fma(a[1], b[0], a[0] * b[1])}; }
inline v2f cmul_conj(v2f a, v2f b) { return v2f{std::fma(a[1], b[1], a[0] * b[0]), std::fma(a[1], b[0], -(a[0] * b[1]))}; }
inline v2f chalf_add_mi(v2f a, v2f b) { return v2f{std::fma(a[0], 0.5f, b[1]), std::fma(a[1], 0.5f, -b[0])}; }
inline v2f chalf_add_pi(v2f a, v2f b) { return v2f{std::fma(a[0], 0.5f, -b[1]), std::fma(a[1], 0.5f, b[0])}; }
inline v2f cconj_half_add_pi(v2f a, v2f b) { return v2f{std::fma(a[0], 0.5f, -b[1]), std::fma(a[1], -0.5f, -b[0])}; }
inline v2f chalf_conj_add_iconj(v2f a, v2f b) { return v2f{std::fma(a[0], 0.5f, b[1]), std::fma(a[1], -0.5f, b[0])}; }
}  // namespace dn
