// Shared by the translation units of libsoccerdiffusion_hip.so (not part of the public ABI).
#ifndef SD_COMMON_H
#define SD_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SD_LN_EPS 1e-5f

// Stores of row tensors that the same kernel does not read again.  SD_PLAIN_STORES=1 builds them as plain stores (A/B runs:
// tools/exp/store_pattern.hip measures non-temporal stores ~8 % slower than plain ones in a pure store stream).
#ifndef SD_PLAIN_STORES
#define SD_PLAIN_STORES 0
#endif
#if SD_PLAIN_STORES
#define SD_NT_STORE(v, p) (*(p) = (v))
#else
#define SD_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

// error reporting (definitions in sd_kernels.hip)
int fail(int code, const char *msg);

// hipGetLastError() is sticky per thread and also reports errors left behind by other
// users of the runtime in this process (e.g. a probe made by the host framework), so
// every launch first clears it and then checks only its own result.
#define SD_CHECK_LAUNCH(name)                                 \
    do {                                                      \
        hipError_t e_ = hipGetLastError();                    \
        if (e_ != hipSuccess) return fail((int)e_, name);     \
    } while (0)
#define SD_LAUNCH(...)                    \
    do {                                  \
        (void)hipGetLastError();          \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)

// "Done on the CURRENT device": hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting and one process may drive
// several devices, so a call site's once-flag is one flag per device.  Used as `static DevFlag attr_set;  if (!attr_set) { ...;
// attr_set = true; }` - a race at worst sets the attribute twice.
struct DevFlag {
    static constexpr int MAX_DEV = 64;
    std::atomic<bool> f[MAX_DEV];
    DevFlag() { for (auto &x : f) x.store(false, std::memory_order_relaxed); }
    static int dev() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEV) d = 0;
        return d;
    }
    bool operator!() const { return !f[dev()].load(std::memory_order_acquire); }
    DevFlag &operator=(bool v) {
        f[dev()].store(v, std::memory_order_release);
        return *this;
    }
};

// --------------------------------------------------------------------------------------
// Optional per-launch timing (bench.py's roofline leg): when enabled, every kernel launch
// is bracketed by a hipEvent pair on the launch stream, tagged with its kernel class.
// Off by default; never enable while capturing a graph (events are created lazily).
// --------------------------------------------------------------------------------------
struct ProfRec { int cls; hipEvent_t a, b; };
struct ProfScope {
    bool on;
    ProfRec rec;
    hipStream_t st;
    ProfScope(int cls, hipStream_t s);
    ~ProfScope();
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// all-reduce sum inside each group of 16 consecutive lanes (one DPP row) with row rotations:
// 4 VALU-rate steps instead of ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));  // row_ror:1
    return v;
}

// 4x4 transpose inside every quad of lanes (lane i = lane & 3 holds column i of rows x0..x3 on
// entry, row i of columns 0..3 on exit; an involution).  Turns the MFMA accumulator's "4
// consecutive rows per lane" into "4 consecutive columns per lane", so tiles move to and
// from memory 16 bytes per lane (4x fewer store/load instructions: +6..11 % on the panel GEMM).
__device__ __forceinline__ void quad_transpose(float &x0, float &x1, float &x2, float &x3, int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
    float ta = o1 ? x0 : x1, tb = o1 ? x2 : x3;
    ta = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ta), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    tb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tb), 0xB1, 0xF, 0xF, false));
    if (o1) { x0 = ta; x2 = tb; } else { x1 = ta; x3 = tb; }
    float tc = o2 ? x0 : x2, td = o2 ? x1 : x3;
    tc = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tc), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    td = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, td), 0x4E, 0xF, 0xF, false));
    if (o2) { x0 = tc; x1 = td; } else { x2 = tc; x3 = td; }
}

__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

// The same function, branch-free: 0.5 u erfc(-u/sqrt 2) with erfc(a >= 0) = t exp(-a^2 + P(t)), t = 1/(1 + a/2) (the Chebyshev
// fit of Numerical Recipes' erfcc, fractional error < 1.2e-7 over the whole range - and no cancellation on the negative
// side, unlike 1 + erf).  libm's erff compiles to a per-element branch with ~60 instructions on both sides; this is ~20.
__device__ __forceinline__ float gelu_erf_fast(float u) {
    const float z = u * 0.70710678118654752440f, a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, a, 1.0f));
    float p = 0.17087277f;
    p = fmaf(p, t, -0.82215223f);
    p = fmaf(p, t, 1.48851587f);
    p = fmaf(p, t, -1.13520398f);
    p = fmaf(p, t, 0.27886807f);
    p = fmaf(p, t, -0.18628806f);
    p = fmaf(p, t, 0.09678418f);
    p = fmaf(p, t, 0.37409196f);
    p = fmaf(p, t, 1.00002368f);
    p = fmaf(p, t, -1.26551223f);
    const float e = t * __builtin_amdgcn_exp2f(fmaf(-a, a, p) * 1.44269504088896340736f);
    return 0.5f * u * (z >= 0.f ? 2.0f - e : e);
}

// d/du gelu(u) = Phi(u) + u phi(u) with the same erfc fit (Phi(u) = 1 - erfc(u / sqrt 2) / 2, mirrored for u < 0)
__device__ __forceinline__ float gelu_grad_fast(float u) {
    const float z = u * 0.70710678118654752440f, a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, a, 1.0f));
    float p = 0.17087277f;
    p = fmaf(p, t, -0.82215223f);
    p = fmaf(p, t, 1.48851587f);
    p = fmaf(p, t, -1.13520398f);
    p = fmaf(p, t, 0.27886807f);
    p = fmaf(p, t, -0.18628806f);
    p = fmaf(p, t, 0.09678418f);
    p = fmaf(p, t, 0.37409196f);
    p = fmaf(p, t, 1.00002368f);
    p = fmaf(p, t, -1.26551223f);
    const float g = __builtin_amdgcn_exp2f(-a * a * 1.44269504088896340736f);     // exp(-u^2 / 2)
    const float e = 0.5f * t * g * __builtin_amdgcn_exp2f(p * 1.44269504088896340736f);   // erfc(a) / 2
    return (z >= 0.f ? 1.0f - e : e) + u * 0.39894228040143267794f * g;
}

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// two values at a time: the polynomial runs on v_pk_fma_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 u) {
    const f32x2 z = u * 0.70710678118654752440f;
    const f32x2 a = {fabsf(z[0]), fabsf(z[1])};
    const f32x2 d = a * 0.5f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = {0.17087277f, 0.17087277f};
    p = p * t + (-0.82215223f);
    p = p * t + 1.48851587f;
    p = p * t + (-1.13520398f);
    p = p * t + 0.27886807f;
    p = p * t + (-0.18628806f);
    p = p * t + 0.09678418f;
    p = p * t + 0.37409196f;
    p = p * t + 1.00002368f;
    p = p * t + (-1.26551223f);
    const f32x2 x = (p - a * a) * 1.44269504088896340736f;
    const f32x2 e = {t[0] * __builtin_amdgcn_exp2f(x[0]), t[1] * __builtin_amdgcn_exp2f(x[1])};
    const f32x2 w = {z[0] >= 0.f ? 2.0f - e[0] : e[0], z[1] >= 0.f ? 2.0f - e[1] : e[1]};
    return u * 0.5f * w;
}

// erf-GELU through Abramowitz & Stegun 7.1.26: erf(z) = 1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) exp(-z^2), t = 1 / (1 + p z),
// z >= 0, |error| <= 1.5e-7 (absolute): gelu(u) = 0.5 u (1 + sign(u) erf(|u| / sqrt 2)).  Seven instructions per value fewer than
// gelu_erf_fast2 (a degree-5 instead of a degree-9 polynomial); absolute error of the GELU <= 0.75e-7 |u| + fp32 rounding (measured
// max 3.3e-7 on [-8, 8] against fp64 erf, tools/exp/gelu_check.py; gelu_erf_fast2: 3.8e-7).  Used by the trajectory step kernel.
__device__ __forceinline__ f32x2 gelu_erf_as2(f32x2 u) {
    const f32x2 z = f32x2{fabsf(u[0]), fabsf(u[1])} * 0.70710678118654752440f;
    const f32x2 d = z * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    p = p * t;
    const f32x2 x = z * z * (-1.44269504088896340736f);
    const f32x2 e = {p[0] * __builtin_amdgcn_exp2f(x[0]), p[1] * __builtin_amdgcn_exp2f(x[1])};   // 1 - erf(z) = erfc(z)
    // 0.5 u (1 + sign(u) (1 - e)) = u - 0.5 u e for u >= 0, 0.5 u e for u < 0
    const f32x2 hu = u * 0.5f;
    return f32x2{u[0] >= 0.f ? fmaf(-hu[0], e[0], u[0]) : hu[0] * e[0], u[1] >= 0.f ? fmaf(-hu[1], e[1], u[1]) : hu[1] * e[1]};
}

// x*scale as an fp16 pair: hi = fp16(x*scale), lo = fp16(x*scale - hi)  (22 mantissa bits)
__device__ __forceinline__ void f16_split4(const f32x4 &x, float scale, f16x4 &h, f16x4 &l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = x[e] * scale;
        h[e] = (f16)v;
        l[e] = (f16)(v - (float)h[e]);
    }
}

// The same split from packed instructions (v_pk_mul_f32, v_cvt_pkrtz_f16_f32, v_pk_add_f32: ~3 VALU per element instead of
// ~7 with the two halves packed afterwards): hi is rounded toward zero, which costs nothing - lo holds the remainder
// (< 1 ulp of hi, 11 bits) - and lo's own truncation leaves 2^-21 relative instead of 2^-22.
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void f16_split4_pk(const f32x4 &x, float scale, f16x4 &h, f16x4 &l) {
    const f32x2 a = f32x2{x[0], x[1]} * scale, b = f32x2{x[2], x[3]} * scale;
    const fp16x2_t ha = __builtin_amdgcn_cvt_pkrtz(a[0], a[1]), hb = __builtin_amdgcn_cvt_pkrtz(b[0], b[1]);
    const f32x2 ra = a - f32x2{(float)ha[0], (float)ha[1]}, rb = b - f32x2{(float)hb[0], (float)hb[1]};
    const fp16x2_t la = __builtin_amdgcn_cvt_pkrtz(ra[0], ra[1]), lb = __builtin_amdgcn_cvt_pkrtz(rb[0], rb[1]);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(f16x4, u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)});
    l = __builtin_bit_cast(f16x4, u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)});
}

// power of two s with max * s in [8192, 16384) from the bits of an abs-max (m = f 2^e, f in [0.5, 1): s = 2^(14 - e)).
// Exponent arithmetic on the bits: frexpf / ldexpf cost ~25 instructions each with their special cases, and this sits in
// per-row / per-slab paths.  Zero, non-finite and < 2^-113 maxima (the scale would not be a normal float) give 1.
__device__ __forceinline__ float f16_scale_from_bits(unsigned maxbits) {
    const unsigned eb = (maxbits >> 23) & 0xffu;           // biased exponent: frexp's e = eb - 126
    const unsigned sbits = (267u - eb) << 23;              // 2^(14 - e) = 2^(140 - eb)
    return (eb >= 14u && eb != 255u) ? __builtin_bit_cast(float, sbits) : 1.0f;
}

// --------------------------------------------------------------------------------------
// Dropout (training only; reference: torch's default p = 0.1 inside nn.TransformerDecoderLayer / EncoderLayer and
// nn.MultiheadAttention, soccer_diffusion/ml/model/decoder.py:26-33 never overrides it).  ONE mask function for every
// kernel: element (row, col) of the logical (rows x width) tensor of dropout site `site` is kept iff word
// (col & 3) of Philox4x32-7(counter = {quad lo, quad hi, site lo, site hi}, key = seed) is >= thresh, with
// quad = (row * ceil4(width) + col) >> 2 - rows are padded to a multiple of 4 columns so that 4 consecutive columns of
// a row always come from one Philox call, whatever the width (attention rows have S = 11 keys).  Nothing is stored: the
// backward kernels regenerate the mask from (seed, site).  Kept values are scaled by 1 / (1 - p).
// --------------------------------------------------------------------------------------
struct DropoutArgs {
    unsigned thresh;      // keep iff word >= thresh;  0 = no dropout
    float scale;          // 1 / (1 - p)
    unsigned seed_lo, seed_hi, site_lo, site_hi;
    const unsigned *epoch;   // device word added to seed_hi at run time (sd_set_dropout_epoch), or NULL
};
extern const unsigned *g_dropout_epoch;   // sd_kernels.hip

static inline DropoutArgs make_dropout(float p, uint64_t seed, uint64_t site) {
    DropoutArgs a;
    double t = (double)p * 4294967296.0;
    a.thresh = p <= 0.f ? 0u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
    a.scale = p <= 0.f ? 1.0f : 1.0f / (1.0f - p);
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    a.site_lo = (unsigned)site; a.site_hi = (unsigned)(site >> 32);
    a.epoch = g_dropout_epoch;
    return a;
}

__device__ __forceinline__ void philox4x32_7(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&w)[4]) {
    // 7 rounds: the smallest round count Random123 (Salmon et al., SC'11) reports as passing BigCrush; cuRAND / torch run
    // 10.  A round is two quarter-rate 64-bit multiplies (~56 cycles per wave): with 10 rounds the masks were 19 % of the
    // fused forward chain of training (tools/exp/chain_stamps.py).
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        // one 64-bit product per multiplier (v_mad_u64_u32) instead of a mul_lo / mul_hi pair: integer multiplies are
        // quarter rate, and they are most of this function
        const unsigned long p0 = (unsigned long)0xD2511F53u * (unsigned long)c0, p1 = (unsigned long)0xCD9E8D57u * (unsigned long)c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}

// multipliers (0 or 1/(1-p)) of the 4 elements of quad `quad` (= padded flat index >> 2)
__device__ __forceinline__ f32x4 dropout_quad(const DropoutArgs &a, unsigned long quad) {
#ifdef SD_ABL_NOPHILOX   // ablation builds: every element kept
    return f32x4{a.scale, a.scale, a.scale, a.scale} + (float)(quad == 0x123456789ul);
#endif
    unsigned w[4];
    philox4x32_7((unsigned)quad, (unsigned)(quad >> 32), a.site_lo, a.site_hi, a.seed_lo, a.seed_hi + (a.epoch ? *a.epoch : 0u), w);
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = w[e] >= a.thresh ? a.scale : 0.f;
    return m;
}

static inline unsigned grid_for(long n, int block = 256) {
    long g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// row-panel linear layer (sd_kernels.hip): out[R,N] = act(LN?(A) W^T + bias) (+res)
int linear(const float *A, const float *W, const float *bias, const float *ln_w, const float *ln_b, const float *res,
           float *out, int R, int N, int d, int act, hipStream_t s, int lda = 0);

#endif
