/* chroma_math.h -- the numeric contract of the photon-propagation path.
 *
 * Every floating-point primitive the propagate path uses that is NOT a single
 * correctly-rounded IEEE-754 operation is defined here, in plain C, as a fixed
 * sequence of IEEE single-precision +, -, *, /, sqrt and fma operations.  The
 * HIP engine (chroma_amd/csrc) and the CPU oracle (oracle/) both include this
 * header and both are compiled with -ffp-contract=off, so the two sides
 * execute the same rounding sequence and agree bit for bit.  An independent
 * build of the oracle against the host libm (oracle: -DORACLE_LIBM) is kept to
 * show that nothing physical depends on these particular polynomials.
 *
 * What this replaces in the reference: CUDA's device math library
 * (logf/expf/sinf/cosf/tanf/asinf/acosf/atan2f as called from
 * chroma/cuda/photon.h:80,142-144,170,176,197-198,207,221,284,301,314,333,348,
 * 417-420,446,451 and chroma/cuda/rotate.h:24-25, chroma/cuda/cx.h:27-35),
 * which the reference compiles with --use_fast_math (chroma/gpu/tools.py:12)
 * and which therefore has no bit-level definition to match.
 *
 * The approximations are the classic single-precision minimax forms (Cephes
 * lineage): relative error <= ~2 ulp over the domains the path uses.
 * tests/test_math_contract.py checks them against libm in double precision.
 *
 * The includer may define CM_FN (e.g. `__host__ __device__ static inline`).
 */
#ifndef CHROMA_MATH_H
#define CHROMA_MATH_H

#include <stdint.h>

#ifndef CM_FN
#define CM_FN static inline
#endif

#define CM_PI_F        3.141592653589793f   /* chroma/cuda/physical_constants.h:7 */
#define CM_SPEED_OF_LIGHT 299.792458f       /* mm/ns, physical_constants.h:5 */

CM_FN uint32_t cm_f2u(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
CM_FN float cm_u2f(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }

CM_FN float cm_nanf(void) { return cm_u2f(0x7fc00000u); }
CM_FN float cm_inff(void) { return cm_u2f(0x7f800000u); }
CM_FN int cm_isnan(float x) { return (cm_f2u(x) & 0x7fffffffu) > 0x7f800000u; }
CM_FN int cm_isfinite(float x) { return (cm_f2u(x) & 0x7f800000u) != 0x7f800000u; }
CM_FN float cm_fabsf(float x) { return cm_u2f(cm_f2u(x) & 0x7fffffffu); }

/* CUDA min()/max() on floats are fminf/fmaxf: a NaN operand yields the other
 * operand (used by intersect_box, chroma/cuda/intersect.h:119-120). */
CM_FN float cm_fminf(float a, float b) { if (cm_isnan(a)) return b; if (cm_isnan(b)) return a; return (b < a) ? b : a; }
CM_FN float cm_fmaxf(float a, float b) { if (cm_isnan(a)) return b; if (cm_isnan(b)) return a; return (b > a) ? b : a; }

/* float -> int as the GPUs do it (CUDA's cast and v_cvt_i32_f32 alike): toward zero, NaN -> 0, saturating.
 * A C cast of NaN or of an out-of-range value is undefined (x86 gives INT_MIN); the table lookups of the path
 * index with such casts, and a photon whose wavelength has become NaN must read the same table entry on
 * both sides instead of crashing the CPU oracle. */
CM_FN int cm_f2i(float x)
{
    if (cm_isnan(x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return (int)x;
}

/* float -> unsigned the same way (cvt.rzi.u32.f32 / v_cvt_u32_f32): NaN and negatives -> 0, saturating */
CM_FN uint32_t cm_f2u32(float x)
{
    if (cm_isnan(x) || x <= 0.0f) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

CM_FN float cm_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
CM_FN float cm_sqrtf(float x) { return __builtin_sqrtf(x); }
CM_FN float cm_roundf(float x) { return __builtin_roundf(x); }   /* half away from zero, exact */
CM_FN float cm_floorf(float x) { return __builtin_floorf(x); }

/* ---- logf ------------------------------------------------------------- */
CM_FN float cm_logf(float x)
{
    uint32_t ix = cm_f2u(x);
    if (ix == 0u || ix == 0x80000000u) return -cm_inff();
    if (ix >> 31) return cm_nanf();
    if (ix >= 0x7f800000u) return x;            /* +inf or nan */
    int e = 0;
    if (ix < 0x00800000u) {                      /* subnormal: scale by 2^23 */
        x = x * 8388608.0f;
        ix = cm_f2u(x);
        e = -23;
    }
    /* x = m * 2^e, m in [0.5, 1) */
    e += (int)(ix >> 23) - 126;
    float m = cm_u2f((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292E-2f;
    p = cm_fmaf(p, m, -1.1514610310E-1f);
    p = cm_fmaf(p, m,  1.1676998740E-1f);
    p = cm_fmaf(p, m, -1.2420140846E-1f);
    p = cm_fmaf(p, m,  1.4249322787E-1f);
    p = cm_fmaf(p, m, -1.6668057665E-1f);
    p = cm_fmaf(p, m,  2.0000714765E-1f);
    p = cm_fmaf(p, m, -2.4999993993E-1f);
    p = cm_fmaf(p, m,  3.3333331174E-1f);
    float y = (p * m) * z;
    float fe = (float)e;
    y = cm_fmaf(-2.12194440e-4f, fe, y);
    y = cm_fmaf(-0.5f, z, y);
    float r = m + y;
    r = cm_fmaf(0.693359375f, fe, r);
    return r;
}

/* ---- expf ------------------------------------------------------------- */
CM_FN float cm_expf(float x)
{
    if (cm_isnan(x)) return x;
    if (x > 88.72283905206835f) return cm_inff();
    if (x < -87.33654475055310f) return 0.0f;   /* results below FLT_MIN flush to 0 */
    /* n = round(x / ln2) */
    float fn = x * 1.44269504088896341f;
    fn = (fn < 0.0f) ? (fn - 0.5f) : (fn + 0.5f);
    int n = (int)fn;
    fn = (float)n;
    float r = cm_fmaf(fn, -0.693359375f, x);
    r = cm_fmaf(fn, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = cm_fmaf(p, r, 1.3981999507E-3f);
    p = cm_fmaf(p, r, 8.3334519073E-3f);
    p = cm_fmaf(p, r, 4.1665795894E-2f);
    p = cm_fmaf(p, r, 1.6666665459E-1f);
    p = cm_fmaf(p, r, 5.0000001201E-1f);
    float y = cm_fmaf(p, z, r) + 1.0f;
    /* y * 2^n, n in [-126, 128]: split so that both factors are normal */
    int n1 = n / 2, n2 = n - n1;
    y = y * cm_u2f((uint32_t)(n1 + 127) << 23);
    y = y * cm_u2f((uint32_t)(n2 + 127) << 23);
    return y;
}

/* ---- sinf / cosf / tanf: octant reduction, valid for |x| < 8192 -------- */
#define CM_FOPI 1.27323954473516f
#define CM_DP1 0.78515625f
#define CM_DP2 2.4187564849853515625e-4f
#define CM_DP3 3.77489497744594108e-8f

CM_FN float cm_sin_poly(float z /* = r*r */, float r)
{
    float p = -1.9515295891E-4f;
    p = cm_fmaf(p, z,  8.3321608736E-3f);
    p = cm_fmaf(p, z, -1.6666654611E-1f);
    return cm_fmaf(p * z, r, r);
}
CM_FN float cm_cos_poly(float z)
{
    float p = 2.443315711809948E-005f;
    p = cm_fmaf(p, z, -1.388731625493765E-003f);
    p = cm_fmaf(p, z,  4.166664568298827E-002f);
    float y = (p * z) * z;
    y = cm_fmaf(-0.5f, z, y);
    return y + 1.0f;
}

/* reduce |x| to r in [-pi/4, pi/4] and octant index j (even) */
CM_FN float cm_trig_reduce(float ax, int *jout)
{
    int j = (int)(CM_FOPI * ax);
    j = (j + 1) & ~1;
    float y = (float)j;
    float r = cm_fmaf(y, -CM_DP1, ax);
    r = cm_fmaf(y, -CM_DP2, r);
    r = cm_fmaf(y, -CM_DP3, r);
    *jout = j;
    return r;
}

CM_FN void cm_sincosf(float x, float *s, float *c)
{
    if (!cm_isfinite(x) || cm_fabsf(x) > 8192.0f) {
        /* outside the contract's domain: the path never produces such
         * angles; a non-finite argument propagates as NaN like libm. */
        *s = cm_nanf(); *c = cm_nanf();
        return;
    }
    int sneg = (x < 0.0f);
    float ax = cm_fabsf(x);
    int j;
    float r = cm_trig_reduce(ax, &j);
    float z = r * r;
    float ps = cm_sin_poly(z, r);
    float pc = cm_cos_poly(z);
    int q = (j >> 1) & 3;    /* quadrant */
    float sv, cv;
    if (q == 0)      { sv = ps;  cv = pc;  }
    else if (q == 1) { sv = pc;  cv = -ps; }
    else if (q == 2) { sv = -ps; cv = -pc; }
    else             { sv = -pc; cv = ps;  }
    *s = sneg ? -sv : sv;
    *c = cv;
}
CM_FN float cm_sinf(float x) { float s, c; cm_sincosf(x, &s, &c); return s; }
CM_FN float cm_cosf(float x) { float s, c; cm_sincosf(x, &s, &c); return c; }

CM_FN float cm_tanf(float x)
{
    if (!cm_isfinite(x) || cm_fabsf(x) > 8192.0f) return cm_nanf();
    int sneg = (x < 0.0f);
    float ax = cm_fabsf(x);
    int j;
    float r = cm_trig_reduce(ax, &j);
    float z = r * r;
    float y;
    if (ax > 1.0e-4f) {
        float p = 9.38540185543E-3f;
        p = cm_fmaf(p, z, 3.11992232697E-3f);
        p = cm_fmaf(p, z, 2.44301354525E-2f);
        p = cm_fmaf(p, z, 5.34112807005E-2f);
        p = cm_fmaf(p, z, 1.33387994085E-1f);
        p = cm_fmaf(p, z, 3.33331568548E-1f);
        y = cm_fmaf(p * z, r, r);
    } else {
        y = r;
    }
    if (j & 2) y = -1.0f / y;
    return sneg ? -y : y;
}

/* ---- asinf / acosf ------------------------------------------------------ */
CM_FN float cm_asinf(float x)
{
    if (cm_isnan(x)) return x;
    int sneg = (x < 0.0f);
    float a = cm_fabsf(x);
    if (a > 1.0f) return cm_nanf();
    float z, r;
    int big = (a > 0.5f);
    if (a < 1.0e-4f) {
        return x;
    }
    if (big) {
        z = 0.5f * (1.0f - a);
        r = cm_sqrtf(z);
    } else {
        r = a;
        z = r * r;
    }
    float p = 4.2163199048E-2f;
    p = cm_fmaf(p, z, 2.4181311049E-2f);
    p = cm_fmaf(p, z, 4.5470025998E-2f);
    p = cm_fmaf(p, z, 7.4953002686E-2f);
    p = cm_fmaf(p, z, 1.6666752422E-1f);
    float y = cm_fmaf(p * z, r, r);
    if (big) {
        y = y + y;
        y = 1.5707963267948966192f - y;
    }
    return sneg ? -y : y;
}

CM_FN float cm_acosf(float x)
{
    if (cm_isnan(x)) return x;
    if (x < -1.0f || x > 1.0f) return cm_nanf();
    if (x < -0.5f)
        return 3.14159265358979323846f - 2.0f * cm_asinf(cm_sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f)
        return 2.0f * cm_asinf(cm_sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - cm_asinf(x);
}

/* ---- atanf / atan2f ------------------------------------------------------ */
CM_FN float cm_atanf(float x)
{
    if (cm_isnan(x)) return x;
    int sneg = (x < 0.0f);
    float a = cm_fabsf(x);
    float y;
    if (a > 2.414213562373095f) {         /* tan(3pi/8) */
        y = 0.78539816339744830962f * 2.0f;
        a = -(1.0f / a);
    } else if (a > 0.4142135623730950f) { /* tan(pi/8) */
        y = 0.78539816339744830962f;
        a = (a - 1.0f) / (a + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = a * a;
    float p = 8.05374449538e-2f;
    p = cm_fmaf(p, z, -1.38776856032E-1f);
    p = cm_fmaf(p, z,  1.99777106478E-1f);
    p = cm_fmaf(p, z, -3.33329491539E-1f);
    y = y + cm_fmaf(p * z, a, a);
    return sneg ? -y : y;
}

CM_FN float cm_atan2f(float y, float x)
{
    if (cm_isnan(x) || cm_isnan(y)) return cm_nanf();
    if (x == 0.0f) {
        if (y > 0.0f) return 1.5707963267948966192f;
        if (y < 0.0f) return -1.5707963267948966192f;
        /* y == +-0: follow C: atan2(+-0, +0) = +-0, atan2(+-0, -0) = +-pi */
        if (cm_f2u(x) >> 31) return (cm_f2u(y) >> 31) ? -3.14159265358979323846f : 3.14159265358979323846f;
        return y;
    }
    if (!cm_isfinite(x) && !cm_isfinite(y)) {
        float q = (x > 0.0f) ? 0.78539816339744830962f : 2.35619449019234492885f;
        return (y < 0.0f) ? -q : q;
    }
    float w = 0.0f;
    if (x < 0.0f) w = (y < 0.0f || (y == 0.0f && (cm_f2u(y) >> 31))) ? -3.14159265358979323846f : 3.14159265358979323846f;
    float t = cm_atanf(y / x);
    return w + t;
}

/* ---- Philox4x32-10 (Salmon et al., SC'11; the generator cuRAND and rocRAND
 * expose as PHILOX4_32_10).  Key = (seed_lo, seed_hi); counter =
 * (block, stream, photon_id_lo, photon_id_hi); stream 0 is the propagation stream, 1 + k the
 * k-th DAQ acquisition.  Draw k of a photon is word (k & 3)
 * of block (k >> 2).  This replaces the reference's one-XORWOW-state-per-
 * thread-slot scheme (chroma/gpu/tools.py:56-84, chroma/cuda/propagate.cu:241,303)
 * with a per-photon stream, see SURVEY.md fact 3. */
#define CM_PHILOX_M0 0xD2511F53u
#define CM_PHILOX_M1 0xCD9E8D57u
#define CM_PHILOX_W0 0x9E3779B9u
#define CM_PHILOX_W1 0xBB67AE85u

CM_FN void cm_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                            uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)CM_PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)CM_PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += CM_PHILOX_W0; k1 += CM_PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* curand_uniform's mapping of a 32-bit word to (0, 1]:
 * x * 2^-32 + 2^-33 in single precision (CUDA toolkit curand_uniform.h,
 * _curand_uniform; call sites: chroma/cuda/random.h:12 and photon.h). */
CM_FN float cm_u32_to_uniform(uint32_t x)
{
    return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

/* Per-photon RNG stream state kept in registers. */
typedef struct {
    uint32_t key0, key1;     /* seed */
    uint32_t id0, id1;       /* global photon id */
    uint32_t counter;        /* number of draws already taken */
    uint32_t stream;         /* Philox counter word 1: 0 = propagation, 1 + k = k-th DAQ acquisition */
    uint32_t buf[4];         /* current block */
    uint32_t buf_block;      /* block index held in buf, 0xffffffff = none */
} cm_rng;

CM_FN void cm_rng_init(cm_rng *r, uint64_t seed, uint64_t photon_id, uint32_t counter)
{
    r->key0 = (uint32_t)seed; r->key1 = (uint32_t)(seed >> 32);
    r->id0 = (uint32_t)photon_id; r->id1 = (uint32_t)(photon_id >> 32);
    r->counter = counter;
    r->stream = 0u;
    r->buf_block = 0xffffffffu;
    r->buf[0] = r->buf[1] = r->buf[2] = r->buf[3] = 0u;
}

CM_FN float cm_rng_uniform(cm_rng *r)
{
    uint32_t blk = r->counter >> 2;
    if (blk != r->buf_block) {
        cm_philox4x32_10(blk, r->stream, r->id0, r->id1, r->key0, r->key1, r->buf);
        r->buf_block = blk;
    }
    uint32_t lane = r->counter & 3u;
    uint32_t w = (lane == 0u) ? r->buf[0] : (lane == 1u) ? r->buf[1] : (lane == 2u) ? r->buf[2] : r->buf[3];
    r->counter++;
    return cm_u32_to_uniform(w);
}

/* A standard normal deviate by Box-Muller from two uniforms of the stream (the cosine branch only):
 * stands in for curand_normal (call site: chroma/cuda/daq.cu:131). */
CM_FN float cm_rng_normal(cm_rng *r)
{
    float u1 = cm_rng_uniform(r);               /* (0, 1]: the logarithm is finite */
    float u2 = cm_rng_uniform(r);
    float s, c;
    cm_sincosf(6.2831855f * u2, &s, &c);
    return cm_sqrtf(-2.0f * cm_logf(u1)) * c;
}

#endif /* CHROMA_MATH_H */
