// Small complex-arithmetic building blocks shared by the HIP kernels and by the CPU lane
// emulator used in the unit tests.  Everything is forward-transform (e^{-2*pi*i*nk/N}).
#pragma once

#if defined(__HIPCC__)
#define SZ_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SZ_HD inline __attribute__((always_inline))
#endif

namespace sz {

// One complex number = one 8-byte vector, so that complex add/sub map to packed f32 ops.
typedef float v2f __attribute__((vector_size(8)));

SZ_HD v2f mk(float re, float im) { v2f r = {re, im}; return r; }
SZ_HD v2f mul_negi(v2f a) { return mk(a[1], -a[0]); }  // a * (-i)
SZ_HD v2f mul_posi(v2f a) { return mk(-a[1], a[0]); }  // a * (+i)
SZ_HD v2f conj(v2f a) { return mk(a[0], -a[1]); }
SZ_HD v2f scale(v2f a, float s) { return mk(a[0] * s, a[1] * s); }

#if defined(__HIP_DEVICE_COMPILE__)
// On the device the operations that swap or negate one half of a complex number are written as
// the packed instruction with its half-select (op_sel) and negate modifiers.  Left to itself the
// compiler builds the swapped operand with a move and an exclusive-or first, and spends two
// packed multiply-adds and a move on the second half of a complex product.  The roundings are
// those of the expressions in the other branch with the product-sum contracted into one fma.
SZ_HD v2f cmul(v2f a, v2f b)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));   // (a.im b.im, a.im b.re)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (a.re b.re - t.lo, a.re b.im + t.hi)
    return r;
}
SZ_HD v2f add_negi(v2f a, v2f b)   // a + (-i) b = (a.re + b.im, a.im - b.re)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SZ_HD v2f sub_negi(v2f a, v2f b)   // a - (-i) b = (a.re - b.im, a.im + b.re)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SZ_HD v2f add_conj(v2f a, v2f b)   // a + conj(b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SZ_HD v2f sub_conj(v2f a, v2f b)   // a - conj(b)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
SZ_HD v2f cmul(v2f a, v2f b) { return mk(a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]); }
SZ_HD v2f add_negi(v2f a, v2f b) { return a + mul_negi(b); }
SZ_HD v2f sub_negi(v2f a, v2f b) { return a - mul_negi(b); }
SZ_HD v2f add_conj(v2f a, v2f b) { return a + conj(b); }
SZ_HD v2f sub_conj(v2f a, v2f b) { return a - conj(b); }
#endif

// 4-point DFT
SZ_HD void dft4(v2f a, v2f b, v2f c, v2f d, v2f& y0, v2f& y1, v2f& y2, v2f& y3)
{
    const v2f s0 = a + c, d0 = a - c, s1 = b + d, d1 = b - d;
    y0 = s0 + s1;
    y2 = s0 - s1;
    y1 = add_negi(d0, d1);
    y3 = sub_negi(d0, d1);
}

// 5-point DFT
SZ_HD void dft5(v2f x0, v2f x1, v2f x2, v2f x3, v2f x4, v2f& y0, v2f& y1, v2f& y2, v2f& y3, v2f& y4)
{
    const float kC = 0.559016994374947424f;   // (cos(2pi/5) - cos(4pi/5)) / 2
    const float kS1 = 0.951056516295153572f;  // sin(2pi/5)
    const float kS2 = 0.587785252292473129f;  // sin(4pi/5)
    const v2f t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    const v2f t5 = t1 + t2;
    y0 = x0 + t5;
    const v2f m1 = x0 - scale(t5, 0.25f);
    const v2f m2 = scale(t1 - t2, kC);
    const v2f s1 = m1 + m2, s2 = m1 - m2;
    const v2f w1 = scale(t3, kS1) + scale(t4, kS2);
    const v2f w2 = scale(t3, kS2) - scale(t4, kS1);
    y1 = add_negi(s1, w1);
    y4 = sub_negi(s1, w1);
    y2 = add_negi(s2, w2);
    y3 = sub_negi(s2, w2);
}

// 20-point DFT by the prime-factor (Good-Thomas) map: n = (5*n1 + 4*n2) mod 20 in,
// k = (5*k1 + 16*k2) mod 20 out; no twiddles between the 4- and 5-point stages.
// All indices are compile-time after unrolling, so x[] and y[] live in registers.
SZ_HD void dft20(const v2f (&x)[20], v2f (&y)[20])
{
    v2f t[4][5];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2)
        dft4(x[(4 * n2) % 20], x[(5 + 4 * n2) % 20], x[(10 + 4 * n2) % 20], x[(15 + 4 * n2) % 20],
             t[0][n2], t[1][n2], t[2][n2], t[3][n2]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
        dft5(t[k1][0], t[k1][1], t[k1][2], t[k1][3], t[k1][4],
             y[(5 * k1) % 20], y[(5 * k1 + 16) % 20], y[(5 * k1 + 32) % 20], y[(5 * k1 + 48) % 20],
             y[(5 * k1 + 64) % 20]);
}

}  // namespace sz
