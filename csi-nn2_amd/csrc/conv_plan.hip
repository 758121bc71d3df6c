// conv_plan.hip -- per-layer plans: algorithm choice, weight packing, epilogue tables, launch.
//
// A plan owns ONE device allocation (the "constant block"):
//     [ packed weights | acc_init (int32[Co]) | mult (float[Co]) | bias (float[Co]) ]
// with 256-byte aligned sections.  The block is position independent, so a multi-GPU caller
// builds it on rank 0 and broadcasts the bytes; other ranks create the plan with
// kernel_host == NULL and receive the block.
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "igemm_common.h"

struct shl_mi355x_conv_plan {
    shl_mi355x_conv_desc desc;
    int algo;
    char *block;       // device
    size_t block_bytes;
    size_t off_w, off_acc, off_mult, off_bias, off_pad;
    size_t off_wfrag;  // 0: absent; pointwise int8 weights in MFMA fragment order (conv1x1_stream.hip)
    size_t off_wpatch; // 0: absent; 3x3 weights as the per-wave fragment streams of conv_igemm_patch.hip
    size_t off_flags;  // 64-byte record of the plan's host-derived epilogue choices (PlanFlags): travels with a broadcast
    int32_t pt_geom;   // wave roles of that kernel, chosen for desc.batch (the packing depends on them)
    // int8 epilogue shortcuts (see ConvArgs)
    int32_t div_exact, div_fma, act_clamp;
    float clamp_lo, clamp_hi, inv_out_scale;
    // NCHW through the NHWC MFMA kernel: scratch images of the input and output, sized for
    // desc.batch at plan time (no allocation may happen inside a captured forward)
    char *scratch_in, *scratch_out;
    int2 *pix_tab;  // per output pixel: tap-(0,0) byte offset and tap validity masks (ConvArgs::pix_tab), or NULL
    float ch_in_scale, ch_out_scale;  // ALGO_DW_CHANNEL: input scale, output scale from multiplier / shift
    int32_t ch_has_bias;
    int32_t kstride;   // igemm: packed row bytes
    int32_t kchunks;
    int32_t cchunks;
    const char *kernel_name;
    const char *variant;  // implicit-GEMM family measured best at plan time (tune_plan), nullptr = the selection rules
    int32_t no_stream_consumer;  // a depthwise layer whose consumer is NOT a pointwise layer dwpw_stream fuses with (set by the graph's owner)
};

namespace shl {

// What the host derived from the tables of the rank that packed the block and the kernels depend on.  Stored at the end
// of the constant block so that a receiver of a weight broadcast adopts the ROOT's choices together with the root's
// tables (a rank initialised with placeholder tables may have derived other ones): shl_mi355x_conv_plan_adopt_block.
struct PlanFlags {
    uint32_t magic;
    int32_t div_exact, div_fma, act_clamp;
    float clamp_lo, clamp_hi, inv_out_scale;
    int32_t pt_geom;
    // what the packing of the block itself depends on: a receiver whose own plan differs holds bytes it cannot read
    int32_t algo, kstride;
    uint64_t block_bytes;
    uint32_t reserved[4];
};
static_assert(sizeof(PlanFlags) == 64, "flags record");
constexpr uint32_t PLAN_FLAGS_MAGIC = 0x53484c46u;  // "SHLF"

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int esize_of(const shl_mi355x_conv_desc &d) { return d.dtype == SHL_MI355X_I8 ? 1 : 2; }

static bool is_depthwise(const shl_mi355x_conv_desc &d) { return d.group > 1 && d.group == d.in_c; }

static int validate(const shl_mi355x_conv_desc &d)
{
    if (d.layout != SHL_MI355X_NHWC && d.layout != SHL_MI355X_NCHW) return SHL_MI355X_EINVAL;
    if (d.dtype != SHL_MI355X_I8 && d.dtype != SHL_MI355X_F16) return SHL_MI355X_EINVAL;
    if (d.act < SHL_MI355X_ACT_NONE || d.act > SHL_MI355X_ACT_RELU6) return SHL_MI355X_EINVAL;
    if (d.batch < 0 || d.in_h <= 0 || d.in_w <= 0 || d.in_c <= 0) return SHL_MI355X_EINVAL;
    if (d.out_h <= 0 || d.out_w <= 0 || d.out_c <= 0) return SHL_MI355X_EINVAL;
    if (d.kernel_h <= 0 || d.kernel_w <= 0) return SHL_MI355X_EINVAL;
    if (d.stride_h <= 0 || d.stride_w <= 0 || d.dilation_h <= 0 || d.dilation_w <= 0)
        return SHL_MI355X_EINVAL;
    if (d.group <= 0 || d.in_c % d.group || d.out_c % d.group) return SHL_MI355X_EINVAL;
    if (d.pad_top < 0 || d.pad_left < 0) return SHL_MI355X_EINVAL;
    // output positions that lie entirely in the bottom / right padding are legal (1-wide kernels
    // with pad_right: the reference and torch produce bias there); every tap is bounds-checked
    if (d.dtype == SHL_MI355X_I8 && !(d.out_scale > 0.0f)) return SHL_MI355X_EINVAL;
    return SHL_MI355X_OK;
}

// ---- host replicas of the literal int8 epilogue (this file is compiled with -ffp-contract=off)
static int host_sat8(float r)
{
    r = r > 127.0f ? 127.0f : r;
    r = r < -128.0f ? -128.0f : r;
    return (int)r;
}

static int host_requant(float x, float s, int zp) { return host_sat8(__builtin_rintf(x / s) + (float)zp); }

// "dequantise, relu(6), requantise with the same record" on a saturated q is a monotone map of
// 256 values; when it equals clamp(q, requant(0), requant(6) | 127) for every q -- checked here
// against the literal formula -- the kernels fold it, together with the int8 saturation, into one
// clamp of r = rint(f / s) + zp.  Returns false when the scale is too exotic for the shortcut
// (the kernels then run the literal code).
static bool derive_act_clamp(const shl_mi355x_conv_desc &d, float *lo, float *hi)
{
    const float s = d.out_scale;
    const int zp = d.out_zp;
    const int qlo = host_requant(0.0f, s, zp);
    const int qhi = d.act == SHL_MI355X_ACT_RELU6 ? host_requant(6.0f, s, zp) : 127;
    if (qlo > qhi) return false;
    for (int q = -128; q <= 127; ++q) {
        float x = ((float)q - (float)zp) * s;
        x = x > 0.0f ? x : 0.0f;
        if (d.act == SHL_MI355X_ACT_RELU6) x = x < 6.0f ? x : 6.0f;
        const int literal = host_requant(x, s, zp);
        const int fast = q < qlo ? qlo : (q > qhi ? qhi : q);
        if (literal != fast) return false;
    }
    *lo = (float)qlo;
    *hi = (float)qhi;
    return true;
}

static bool is_pow2_scale(float s)
{
    int e;
    return s > 0.0f && __builtin_frexpf(s, &e) == 0.5f && e > -40 && e < 40;
}

// A power-of-two output scale is folded into the multiplier and bias tables: fl(fl(S m) + b) / s ==
// fl(fl(S (m / s)) + b / s) as long as no operand leaves the normal range, which these bounds guarantee
// (scaling by a power of two commutes with rounding).  Two packed multiplications less per four outputs.
static bool pow2_fold_ok(const shl_mi355x_conv_desc &d, const float *mult_host, const float *bias_host)
{
    if (d.dtype != SHL_MI355X_I8 || !is_pow2_scale(d.out_scale)) return false;
    for (int oc = 0; oc < d.out_c; ++oc) {
        const float m = fabsf(mult_host ? mult_host[oc] : 1.0f), b = fabsf(bias_host ? bias_host[oc] : 0.0f);
        if (!((m == 0.0f || (m >= 0x1p-60f && m <= 0x1p28f)) && (b == 0.0f || (b >= 0x1p-60f && b <= 0x1p60f)))) return false;
    }
    return true;
}

// Division by the output scale as multiply + two fma corrections (common.h div_by_scale; y = RN(1/s) is the
// plan's inv_out_scale, one IEEE division on the host): no overflow or underflow can touch a quotient that
// matters when 2^-40 <= s <= 2^40 and |fl(fl(S m) + b)| <= 2^60 for every int32 S (|q| < 2^-20 rounds to 0
// whatever its low bits are).
static bool fma_division_ok(const shl_mi355x_conv_desc &d, const float *mult_host, const float *bias_host)
{
    if (d.dtype != SHL_MI355X_I8 || !(d.out_scale >= 0x1p-40f && d.out_scale <= 0x1p40f)) return false;
    for (int oc = 0; oc < d.out_c; ++oc) {
        const float bound = 4294967296.0f * fabsf(mult_host ? mult_host[oc] : 1.0f) + fabsf(bias_host ? bias_host[oc] : 0.0f);
        if (!(bound <= 0x1p60f)) return false;  // also NaN
    }
    return true;
}

static int choose_algo(const shl_mi355x_conv_desc &d)
{
    if (is_depthwise(d)) return (dwconv_supports(d) || dwconv_nchw_supports(d)) ? SHL_MI355X_ALGO_DW : SHL_MI355X_ALGO_DIRECT;
    if (d.group != 1) return -1;  // grouped convolution: SURVEY 8f3
    if (stem_supports(d)) return SHL_MI355X_ALGO_STEM;
    return igemm_supports(d) ? SHL_MI355X_ALGO_IGEMM : SHL_MI355X_ALGO_DIRECT;
}

// weights -> [Co][kstride] rows in (ky, kx, ic) order, zero padded (igemm)
static void pack_igemm(const shl_mi355x_conv_desc &d, const char *src, char *dst, int kstride)
{
    const int es = esize_of(d);
    const int K = d.kernel_h * d.kernel_w * d.in_c;
    memset(dst, 0, (size_t)d.out_c * kstride);
    for (int oc = 0; oc < d.out_c; ++oc) {
        char *row = dst + (size_t)oc * kstride;
        if (d.layout == SHL_MI355X_NHWC) {  // OHWI is already K-major
            memcpy(row, src + (size_t)oc * K * es, (size_t)K * es);
        } else {  // OIHW
            for (int ic = 0; ic < d.in_c; ++ic)
                for (int ky = 0; ky < d.kernel_h; ++ky)
                    for (int kx = 0; kx < d.kernel_w; ++kx) {
                        const size_t s = (((size_t)oc * d.in_c + ic) * d.kernel_h + ky) * d.kernel_w + kx;
                        const size_t k = ((size_t)ky * d.kernel_w + kx) * d.in_c + ic;
                        memcpy(row + k * es, src + s * es, es);
                    }
        }
    }
}

// ---- plan-time choice of the implicit-GEMM family by MEASUREMENT ---------------------------------
// The selection rules of conv_igemm.hip:igemm_pick are threshold ladders tuned at batch 1 and 128 on two networks; in
// between they were wrong by up to 2x (profiles/r03_notes.md).  So a plan whose layer the implicit-GEMM kernels take
// times the rules' own pick against every family forced in turn -- wave, tile, ping-pong, producer / consumer, row-patch;
// a family that does not apply resolves to "tile", i.e. costs one redundant timing -- on scratch tensors of the plan's
// own shape and batch, and keeps a challenger only if it beats the rules by more than 3 %.  Results are cached per
// (shape, layout, dtype, batch, epilogue flavour) for the life of the process (ResNet-50's 16 3x3 layers are 7 shapes).
// All families are exact in int8 and within the same 1e-3 in binary16, so the choice never changes results.
// SHL_MI355X_TUNE=0 turns it off; SHL_MI355X_IGEMM=<family> (A/B runs, the forced-variant tests) implies that.
// Default: int8 plans only.  Every int8 family is bit-exact (int32 accumulation, one epilogue), so a pick that flips with
// timing noise can never change a result.  The binary16 families sum fp32 in different orders: two processes (or two
// ranks) with identical inputs could then differ in the last bit, so binary16 plans are tuned only on request
// (SHL_MI355X_TUNE=1 or "all") and keep the selection rules otherwise.
static bool tuning_enabled(int dtype)
{
    const char *e = getenv("SHL_MI355X_TUNE");  // read per plan: a caller may switch tuning off for some layers
    if (e && e[0] == '0') return false;
    if (igemm_env_override()) return false;
    if (dtype != SHL_MI355X_I8) return e && (e[0] == '1' || e[0] == 'a');
    return true;
}

static const char *family_kernel_name(const char *v, bool i8)
{
    if (!strcmp(v, "gemv")) return i8 ? "conv_gemv_i8_dot4" : "conv_gemv_f16_fma";
    if (!strcmp(v, "regs")) return i8 ? "conv_igemm_regs_i8_mfma32x32x32" : "conv_igemm_regs_f16_mfma32x32x16";
    if (!strcmp(v, "stream1x1")) return "conv1x1_stream_i8_mfma32x32x32";
    if (!strcmp(v, "resident1x1")) return "conv1x1_resident_i8_mfma32x32x32";
    if (!strcmp(v, "nchw1x1")) return i8 ? "conv1x1_nchw_i8" : "conv1x1_nchw_f16";
    if (!strcmp(v, "wave")) return i8 ? "conv_igemm_wave_i8_mfma32x32x32" : "conv_igemm_wave_f16_mfma32x32x16";
    if (!strcmp(v, "tile")) return i8 ? "conv_igemm_tile_i8_mfma32x32x32" : "conv_igemm_tile_f16_mfma32x32x16";
    if (!strcmp(v, "pp")) return i8 ? "conv_igemm_pp_i8_mfma32x32x32" : "conv_igemm_pp_f16_mfma32x32x16";
    if (!strcmp(v, "pc")) return i8 ? "conv_igemm_pc_i8_mfma32x32x32" : "conv_igemm_pc_f16_mfma32x32x16";
    return i8 ? "conv_igemm_patch_i8_mfma32x32x32" : "conv_igemm_patch_f16_mfma32x32x16";
}

struct TuneResult {
    const char *variant;  // nullptr: the rules
};
static std::mutex g_tune_lock;
static std::map<std::vector<int32_t>, TuneResult> g_tune_cache;

// the per-pixel address table (8 bytes x N Ho Wo: ~60 MB over ResNet-50's 3x3 layers at batch 128) exists during tuning
// so that the producer / consumer family can be a candidate; it stays only when that family (or a rule that needs the
// table) won
static void release_unused_pix_tab(shl_mi355x_conv_plan *p, bool rules_need_it)
{
    const bool keep = p->variant ? !strcmp(p->variant, "pc") : rules_need_it;
    if (!keep && p->pix_tab) {
        (void)hipFree(p->pix_tab);
        p->pix_tab = nullptr;
    }
}

static std::vector<int32_t> tune_key(const shl_mi355x_conv_plan *p)
{
    const shl_mi355x_conv_desc &d = p->desc;
    return {d.layout, d.dtype, d.act, d.batch, d.in_h, d.in_w, d.in_c, d.out_h, d.out_w, d.out_c,
            d.kernel_h, d.kernel_w, d.stride_h, d.stride_w, d.pad_top, d.pad_left, d.dilation_h,
            d.dilation_w, p->div_exact, p->div_fma, p->act_clamp, p->pt_geom};
}

// a cached verdict for the plan's shape that does not need the per-pixel table: the table (8 B x N Ho Wo, built on the host
// and uploaded) need not exist at all -- it used to be built, uploaded and freed again on every cache hit.
// Limitation, by design: a plan whose table was released runs a LATER forward with a smaller batch on whatever the rules
// pick without the table (the producer / consumer family is then not a candidate) -- correct, possibly slower than a plan
// created for that batch.
static bool tune_cached_without_pix_tab(const shl_mi355x_conv_plan *p)
{
    std::lock_guard<std::mutex> g(g_tune_lock);
    auto it = g_tune_cache.find(tune_key(p));
    return it != g_tune_cache.end() && !(it->second.variant && !strcmp(it->second.variant, "pc"));
}

static void tune_plan(shl_mi355x_conv_plan *p, hipStream_t stream, bool rules_need_pix_tab)
{
    const shl_mi355x_conv_desc &d = p->desc;
    if (!tuning_enabled(d.dtype) || d.batch <= 0) return;
    const bool i8 = d.dtype == SHL_MI355X_I8;
    const std::vector<int32_t> key = tune_key(p);
    {
        std::lock_guard<std::mutex> g(g_tune_lock);
        auto it = g_tune_cache.find(key);
        if (it != g_tune_cache.end()) {
            p->variant = it->second.variant;
            if (p->variant) p->kernel_name = family_kernel_name(p->variant, i8);
            release_unused_pix_tab(p, rules_need_pix_tab);
            return;
        }
    }
    const size_t es = i8 ? 1 : 2;
    const size_t in_b = (size_t)d.batch * d.in_c * d.in_h * d.in_w * es, out_b = (size_t)d.batch * d.out_c * d.out_h * d.out_w * es;
    char *in = nullptr, *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    TuneResult best = {nullptr};
    if (hipMalloc((void **)&in, in_b ? in_b : 16) == hipSuccess && hipMalloc((void **)&out, out_b ? out_b : 16) == hipSuccess &&
        hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
        hipMemsetAsync(in, i8 ? (d.in_zp & 0xff) : 0, in_b, stream) == hipSuccess) {
        static const char *const cands[] = {nullptr, "wave", "tile", "pp", "pc", "patch"};
        float t_rules = 0.f, t_best = 0.f;
        const char *timed[8] = {};  // families already timed (a forced family that does not take the shape resolves to another)
        int ntimed = 0;
        for (const char *c : cands) {
            if (c && !strcmp(c, "patch") && !p->off_wpatch) continue;  // no row-patch weight copy: the family cannot run
            if (c && !strcmp(c, "pc") && !p->pix_tab) continue;
            p->variant = c;
            // two warm launches (code, LDS opt-in, clocks), then the better of two windows of TUNE_REPS back-to-back launches:
            // with three launches per window a 6-us kernel was timed at the host's launch rate and the verdict flipped from
            // run to run (profiles/r04_batch_sweep.txt, batch 2)
            constexpr int TUNE_REPS = 8;
            float ms = -1.f;
            igemm_note_family("");
            bool ok = shl_mi355x_conv_forward(p, in, out, d.batch, stream) == SHL_MI355X_OK &&
                      shl_mi355x_conv_forward(p, in, out, d.batch, stream) == SHL_MI355X_OK;
            // what the forced family resolved to: timing the same kernel twice lets noise alone "win" by 3 %, and the plan would
            // then carry the name of a family that never ran
            const char *resolved = igemm_last_family();
            bool dup = false;
            for (int k = 0; k < ntimed; ++k) dup = dup || !strcmp(timed[k], resolved);
            if (ok && c && (dup || strcmp(resolved, c))) {
                (void)hipStreamSynchronize(stream);
                continue;
            }
            if (ok && ntimed < 8) timed[ntimed++] = resolved;
            for (int w = 0; ok && w < 2; ++w) {
                float t = -1.f;
                ok = hipEventRecord(e0, stream) == hipSuccess;
                for (int r = 0; ok && r < TUNE_REPS; ++r) ok = shl_mi355x_conv_forward(p, in, out, d.batch, stream) == SHL_MI355X_OK;
                if (ok) ok = hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                             hipEventElapsedTime(&t, e0, e1) == hipSuccess;
                if (ok) ms = (ms < 0.f || t < ms) ? t : ms;
            }
            if (!ok) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(stream);
                continue;
            }
            if (!c) t_rules = t_best = ms;
            else if (ms < t_best * 0.97f && ms < t_rules * 0.97f) t_best = ms, best.variant = c;
            static const char *dbg = getenv("SHL_MI355X_DEBUG_TUNE");
            if (dbg) fprintf(stderr, "tune %dx%dx%d->%d k%d s%d b%d %s: %-6s %.2f us\n", d.in_h, d.in_w, d.in_c, d.out_c, d.kernel_h,
                             d.stride_h, d.batch, d.layout == SHL_MI355X_NCHW ? "NCHW" : "NHWC", c ? c : "rules", ms * 1e3f / TUNE_REPS);
        }
        if (t_rules <= 0.f) best.variant = nullptr;  // the rules' own pick could not be timed: keep it
    }
    (void)hipStreamSynchronize(stream);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(in);
    (void)hipFree(out);
    (void)hipGetLastError();
    p->variant = best.variant;
    if (p->variant) p->kernel_name = family_kernel_name(p->variant, i8);
    release_unused_pix_tab(p, rules_need_pix_tab);
    std::lock_guard<std::mutex> g(g_tune_lock);
    g_tune_cache[key] = best;
}

}  // namespace shl

using namespace shl;

extern "C" {

static int plan_create_impl(const struct shl_mi355x_conv_desc *desc, const void *kernel_host, const float *mult_host,
                            const float *bias_host, const int32_t *kernel_zp, void *stream, shl_mi355x_conv_plan **plan_out);

int shl_mi355x_conv_plan_create(const struct shl_mi355x_conv_desc *desc, const void *kernel_host,
                                const float *mult_host, const float *bias_host, void *stream,
                                shl_mi355x_conv_plan **plan_out)
{
    return plan_create_impl(desc, kernel_host, mult_host, bias_host, nullptr, stream, plan_out);
}

int shl_mi355x_conv_plan_create_wzp(const struct shl_mi355x_conv_desc *desc, const void *kernel_host,
                                    const float *mult_host, const float *bias_host, const int32_t *kernel_zp,
                                    void *stream, shl_mi355x_conv_plan **plan_out)
{
    return plan_create_impl(desc, kernel_host, mult_host, bias_host, kernel_zp, stream, plan_out);
}

static int plan_create_impl(const struct shl_mi355x_conv_desc *desc, const void *kernel_host, const float *mult_host,
                            const float *bias_host, const int32_t *kernel_zp, void *stream, shl_mi355x_conv_plan **plan_out)
{
    if (!desc || !plan_out) {
        set_error("conv_plan_create: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    *plan_out = nullptr;
    int rc = validate(*desc);
    if (rc != SHL_MI355X_OK) {
        set_error("conv_plan_create: invalid descriptor");
        return rc;
    }
    const shl_mi355x_conv_desc &d = *desc;
    if (d.dtype == SHL_MI355X_I8 && kernel_host && !mult_host) {
        set_error("conv_plan_create: int8 needs the per-channel multiplier table");
        return SHL_MI355X_EINVAL;
    }
    // int8 output scales: a power of two, or a converter's scale in the range div_by_scale handles -- anything else
    // (out_scale or multipliers beyond 2^+-40) keeps the hardware's division, which only the direct kernel carries
    const bool i8_div_exact = pow2_fold_ok(d, mult_host, bias_host);
    const bool i8_div_fma = !i8_div_exact && fma_division_ok(d, mult_host, bias_host);
    const bool fast_epilogue_ok = d.dtype != SHL_MI355X_I8 || i8_div_exact || i8_div_fma;
    // asymmetric int8 weights (a non-zero kernel zero point in any record): sum (q - zp_in) (w - zp_k[oc]) over the
    // in-image taps, which only the one-output-per-thread kernels compute (the MFMA / depthwise kernels fold zp_in into
    // a per-channel constant and would need a per-PIXEL sum of the receptive field on top)
    bool w_asym = false;
    if (kernel_zp && d.dtype == SHL_MI355X_I8)
        for (int oc = 0; oc < d.out_c; ++oc) w_asym |= kernel_zp[oc] != 0;
    int algo = d.algo;
    if (algo == SHL_MI355X_ALGO_AUTO) {
        algo = choose_algo(d);
        if (algo >= 0 && (!fast_epilogue_ok || w_asym)) algo = SHL_MI355X_ALGO_DIRECT;
    }
    if (w_asym && algo >= 0 && algo != SHL_MI355X_ALGO_DIRECT && algo != SHL_MI355X_ALGO_GROUP) {
        set_error("conv_plan_create: asymmetric weights run on the DIRECT / GROUP kernels only");
        return SHL_MI355X_ENOTSUP;
    }
    if (algo < 0) {
        set_error("conv_plan_create: grouped convolution (group=%d) is not supported", d.group);
        return SHL_MI355X_ENOTSUP;
    }
    // (the grouped direct kernel carries the same hardware-division epilogue as the direct one)
    if (algo >= 0 && algo != SHL_MI355X_ALGO_DIRECT && algo != SHL_MI355X_ALGO_GROUP && !fast_epilogue_ok) {
        set_error("conv_plan_create: only the DIRECT kernel takes an output scale (or multipliers) outside 2^-40 .. 2^40");
        return SHL_MI355X_ENOTSUP;
    }
    if (algo == SHL_MI355X_ALGO_IGEMM && !igemm_supports(d)) {
        set_error("conv_plan_create: IGEMM needs group==1, NHWC and C*esize %% 16 == 0");
        return SHL_MI355X_ENOTSUP;
    }
    if (algo == SHL_MI355X_ALGO_DW && !dwconv_supports(d) && !dwconv_nchw_supports(d)) {
        set_error("conv_plan_create: DW kernel needs NHWC, multiplier 1 and C %% 4 == 0");
        return SHL_MI355X_ENOTSUP;
    }
    if (algo == SHL_MI355X_ALGO_STEM && !stem_supports(d)) {
        set_error("conv_plan_create: STEM kernel needs int8 NHWC 3x3 with 3 input channels, Cout <= 64");
        return SHL_MI355X_ENOTSUP;
    }
    if (algo == SHL_MI355X_ALGO_GROUP && d.group < 2) {
        set_error("conv_plan_create: ALGO_GROUP needs group > 1");
        return SHL_MI355X_EINVAL;
    }
    if (algo != SHL_MI355X_ALGO_GROUP && d.group != 1 && !is_depthwise(d)) {
        set_error("conv_plan_create: grouped convolution (group=%d) runs as SHL_MI355X_ALGO_GROUP only", d.group);
        return SHL_MI355X_ENOTSUP;
    }
    if (algo != SHL_MI355X_ALGO_IGEMM && algo != SHL_MI355X_ALGO_DW && algo != SHL_MI355X_ALGO_DIRECT &&
        algo != SHL_MI355X_ALGO_STEM && algo != SHL_MI355X_ALGO_GROUP) {
        set_error("conv_plan_create: unknown algorithm %d", algo);
        return SHL_MI355X_EINVAL;
    }

    shl_mi355x_conv_plan *p = (shl_mi355x_conv_plan *)calloc(1, sizeof(*p));
    if (!p) return SHL_MI355X_ENOMEM;
    p->desc = d;
    p->algo = algo;
    const int es = esize_of(d);
    const int cpg = d.in_c / d.group;
    const size_t raw_w = (size_t)d.out_c * cpg * d.kernel_h * d.kernel_w * es;
    size_t w_bytes = raw_w;
    bool want_pix_tab = false;
    if (algo == SHL_MI355X_ALGO_IGEMM) {
        const int Kb = d.kernel_h * d.kernel_w * d.in_c * es;
        p->kstride = (int32_t)align_up((size_t)Kb, 64);
        p->kchunks = Kb / 16;
        p->cchunks = d.in_c * es / 16;
        w_bytes = (size_t)d.out_c * p->kstride;
        const bool i8 = d.dtype == SHL_MI355X_I8;
        ConvArgs probe = {};
        probe.pix_tab = reinterpret_cast<const int2 *>(&probe);  // "will exist": built below when the pick needs it
        p->pt_geom = patch_choose_geom(d, d.batch);
        if (p->pt_geom) {
            probe.w_patch = &probe;  // "will exist"
            probe.pt_geom = p->pt_geom;
            probe.in_nchw = d.layout == SHL_MI355X_NCHW;
            probe.act = d.act;
            float lo, hi;
            probe.act_clamp = d.act != SHL_MI355X_ACT_NONE && derive_act_clamp(d, &lo, &hi);
        }
        probe.Kh = d.kernel_h, probe.Kw = d.kernel_w, probe.sh = d.stride_h, probe.sw = d.stride_w;
        probe.dh = d.dilation_h, probe.dw = d.dilation_w;
        probe.pt = d.pad_top, probe.pl = d.pad_left, probe.H = d.in_h, probe.W = d.in_w, probe.Ho = d.out_h, probe.Wo = d.out_w;
        probe.C = d.in_c, probe.Co = d.out_c, probe.kstride = p->kstride;
        probe.M = (int32_t)((int64_t)d.batch * d.out_h * d.out_w);
        probe.N = d.batch;
        probe.out_nchw = d.layout == SHL_MI355X_NCHW && ((d.out_h * d.out_w * es) & 3) == 0;
        // as shl_mi355x_conv_forward decides: an NCHW layer first asks the NCHW-native row-patch kernel (input AND output
        // NCHW), everything else sees the NHWC view of the re-layout path
        const char *v = nullptr;
        bool patch_native_nchw = false;
        if (probe.in_nchw) {
            ConvArgs t = probe;
            t.out_nchw = 1;
            if (!strcmp(igemm_pick_name(t, es), "patch")) v = "patch", patch_native_nchw = true;
            probe.in_nchw = 0;
        }
        if (!v && es == 2 && d.layout == SHL_MI355X_NCHW && probe.w_patch) {  // binary16 NCHW: the row-patch kernel on the NHWC view (conv_forward)
            ConvArgs t = probe;
            t.out_nchw = 0;
            if (!strcmp(igemm_pick_name(t, es), "patch")) v = "patch";
        }
        if (!v) v = igemm_pick_name(probe, es);
        if (!strcmp(v, "wave"))
            p->kernel_name = i8 ? "conv_igemm_wave_i8_mfma32x32x32" : "conv_igemm_wave_f16_mfma32x32x16";
        else if (!strcmp(v, "regs"))
            p->kernel_name = i8 ? "conv_igemm_regs_i8_mfma32x32x32" : "conv_igemm_regs_f16_mfma32x32x16";
        else if (!strcmp(v, "gemv"))
            p->kernel_name = i8 ? "conv_gemv_i8_dot4" : "conv_gemv_f16_fma";
        else if (!strcmp(v, "patch"))
            p->kernel_name = i8 ? "conv_igemm_patch_i8_mfma32x32x32"
                                : (patch_native_nchw ? "conv_igemm_patch_nchw_f16_mfma32x32x16" : "conv_igemm_patch_f16_mfma32x32x16");
        else if (!strcmp(v, "pp"))
            p->kernel_name = i8 ? "conv_igemm_pp_i8_mfma32x32x32" : "conv_igemm_pp_f16_mfma32x32x16";
        else if (!strcmp(v, "res")) {
            p->kernel_name = "conv_igemm_res_i8_mfma32x32x32";
            want_pix_tab = true;
        } else if (!strcmp(v, "pc")) {
            p->kernel_name = i8 ? "conv_igemm_pc_i8_mfma32x32x32" : "conv_igemm_pc_f16_mfma32x32x16";
            want_pix_tab = true;
        }
        else
            p->kernel_name = i8 ? "conv_igemm_tile_i8_mfma32x32x32" : "conv_igemm_tile_f16_mfma32x32x16";
        if (i8 && d.layout == SHL_MI355X_NHWC && !igemm_env_override()) {  // pointwise forms of their own (launch_conv_igemm asks them first unless a family is forced)
            probe.w_frag = &probe;  // the copy is made below for exactly these shapes
            probe.div_exact = i8_div_exact, probe.div_fma = i8_div_fma, probe.act = d.act;
            {
                float lo, hi;
                probe.act_clamp = d.act != SHL_MI355X_ACT_NONE && derive_act_clamp(d, &lo, &hi);
            }
            if (conv1x1_stream_pick(probe)) p->kernel_name = "conv1x1_stream_i8_mfma32x32x32";
            if (conv1x1_resident_pick(probe)) p->kernel_name = "conv1x1_resident_i8_mfma32x32x32";
            if (conv1x1_latency_pick(probe)) p->kernel_name = "conv1x1_latency_i8_mfma32x32x32";
        }
    } else if (algo == SHL_MI355X_ALGO_STEM) {
        w_bytes = stem_weight_bytes(d);
        p->kernel_name = "conv_stem_i8_dot4";
        {  // the plan's batch decides the name (conv_stem.hip:launch_conv_stem picks per call)
            ConvArgs t = {};
            t.N = d.batch, t.H = d.in_h, t.W = d.in_w, t.C = d.in_c, t.Co = d.out_c, t.dh = d.dilation_h, t.dw = d.dilation_w;
            t.M = (int32_t)((int64_t)d.batch * d.out_h * d.out_w);
            if (stem_mfma_pick(t)) p->kernel_name = "conv_stem_i8_mfma32x32x32";
        }
    } else if (algo == SHL_MI355X_ALGO_DW) {
        p->kernel_name = d.dtype == SHL_MI355X_I8 ? "dwconv_nhwc_i8" : "dwconv_nhwc_f16";
        if (d.layout == SHL_MI355X_NCHW) p->kernel_name = d.dtype == SHL_MI355X_I8 ? "dwconv3x3_nchw_i8" : "dwconv3x3_nchw_f16";
        if (dwconv_dot4_supports(d)) {
            p->kstride = 12;  // marks the [C][3 dwords] packing for launch_dwconv
            w_bytes = (size_t)d.in_c * 12;
            if (d.dilation_h == 1 && d.dilation_w == 1 &&
                dwconv_mfma_pick((int64_t)d.batch * d.out_h * d.out_w, d.in_c, d.in_h, d.in_w, d.out_h, d.out_w,
                                 d.stride_h, d.stride_w))
                p->kernel_name = "dwconv_mfma_i8";
        }
    } else if (algo == SHL_MI355X_ALGO_GROUP) {
        p->kernel_name = d.dtype == SHL_MI355X_I8 ? (w_asym ? "conv_group_direct_i8_wzp" : "conv_group_direct_i8") : "conv_group_direct_f16";
    } else {
        p->kernel_name = d.dtype == SHL_MI355X_I8 ? (w_asym ? "conv_direct_i8_wzp" : "conv_direct_i8") : "conv_direct_f16";
    }
    // tables are padded to a multiple of 128 channels so that kernels may fetch whole tile rows
    const size_t tab_bytes = align_up((size_t)d.out_c, 128) * 4;
    p->off_w = 0;
    p->off_acc = align_up(w_bytes, 256);
    p->off_mult = p->off_acc + tab_bytes;
    p->off_bias = p->off_mult + tab_bytes;
    p->off_pad = p->off_bias + tab_bytes;
    p->block_bytes = p->off_pad + PAD_PAGE_BYTES;
    // pointwise layers whose K row is a multiple of 32 bytes: a second copy of the weights in fragment order
    // [32-channel group][K / 32][64 lanes][16 B] so that a wave fetches an A fragment with one coalesced
    // 1 KiB load (from [Cout][K] rows it is 64 lines per load instruction)
    const bool frag_copy = algo == SHL_MI355X_ALGO_IGEMM && d.kernel_h == 1 && d.kernel_w == 1 &&
                           (d.in_c * es) % 32 == 0 && d.in_c * es <= 4096 && d.out_c % 32 == 0;
    if (frag_copy) {
        p->off_wfrag = p->block_bytes;
        p->block_bytes += (size_t)d.out_c * d.in_c * es;
    }
    if (algo == SHL_MI355X_ALGO_IGEMM && p->pt_geom) {
        p->off_wpatch = align_up(p->block_bytes, 256);
        p->block_bytes = p->off_wpatch + patch_weight_bytes(d, p->pt_geom);
    }
    p->off_flags = align_up(p->block_bytes, 256);
    p->block_bytes = p->off_flags + sizeof(PlanFlags);
    p->inv_out_scale = 1.0f / d.out_scale;
    if (d.dtype == SHL_MI355X_I8) {
        p->div_exact = i8_div_exact;
        p->div_fma = i8_div_fma;
        p->clamp_lo = -128.0f;
        p->clamp_hi = 127.0f;
        p->act_clamp = d.act != SHL_MI355X_ACT_NONE && derive_act_clamp(d, &p->clamp_lo, &p->clamp_hi);
    }

    hipError_t e = hipMalloc((void **)&p->block, p->block_bytes);
    if (e != hipSuccess) {
        free(p);
        return hip_fail(e, "hipMalloc(plan block)");
    }

    std::vector<char> host(p->block_bytes, 0);
    {
        PlanFlags f = {};
        f.magic = PLAN_FLAGS_MAGIC;
        f.div_exact = p->div_exact, f.div_fma = p->div_fma, f.act_clamp = p->act_clamp;
        f.clamp_lo = p->clamp_lo, f.clamp_hi = p->clamp_hi, f.inv_out_scale = p->inv_out_scale;
        f.pt_geom = p->pt_geom;
        f.algo = algo, f.kstride = p->kstride, f.block_bytes = p->block_bytes;
        memcpy(host.data() + p->off_flags, &f, sizeof(f));
    }
    // padding value: the input zero point (int8) / 0.0 (f16)
    memset(host.data() + p->off_pad, d.dtype == SHL_MI355X_I8 ? (d.in_zp & 0xFF) : 0, PAD_PAGE_BYTES);
    if (kernel_host) {
        const char *src = static_cast<const char *>(kernel_host);
        if (algo == SHL_MI355X_ALGO_IGEMM) {
            pack_igemm(d, src, host.data() + p->off_w, p->kstride);
            if (frag_copy) {
                char *dst = host.data() + p->off_wfrag;
                for (int g = 0; g < d.out_c / 32; ++g)
                    for (int sub = 0; sub < d.in_c * es / 32; ++sub)
                        for (int lane = 0; lane < 64; ++lane, dst += 16)
                            memcpy(dst, host.data() + p->off_w + (size_t)(g * 32 + (lane & 31)) * p->kstride + sub * 32 + (lane >> 5) * 16, 16);
            }
            if (p->off_wpatch)
                patch_pack_weights(d, p->pt_geom, reinterpret_cast<const int8_t *>(src),
                                   reinterpret_cast<int8_t *>(host.data() + p->off_wpatch));
        }
        else if (algo == SHL_MI355X_ALGO_STEM)
            stem_pack_weights(d, reinterpret_cast<const int8_t *>(src),
                              reinterpret_cast<int32_t *>(host.data() + p->off_w));
        else if (algo == SHL_MI355X_ALGO_DW && p->kstride == 12)
            dwconv_dot4_pack(d, reinterpret_cast<const int8_t *>(src),
                             reinterpret_cast<uint32_t *>(host.data() + p->off_w));
        else
            memcpy(host.data() + p->off_w, src, raw_w);
        int32_t *acc = reinterpret_cast<int32_t *>(host.data() + p->off_acc);
        float *mult = reinterpret_cast<float *>(host.data() + p->off_mult);
        float *bias = reinterpret_cast<float *>(host.data() + p->off_bias);
        for (int oc = 0; oc < d.out_c; ++oc) {
            mult[oc] = mult_host ? mult_host[oc] : 1.0f;
            bias[oc] = bias_host ? bias_host[oc] : 0.0f;
            if (p->div_exact) {  // see pow2_fold_ok
                mult[oc] *= p->inv_out_scale;
                bias[oc] *= p->inv_out_scale;
            }
            acc[oc] = w_asym ? kernel_zp[oc] : 0;  // DIRECT / GROUP: the kernel's zero point per output channel
        }
        if (algo == SHL_MI355X_ALGO_STEM) {
            // padding is materialised as zp_in here too
            const int8_t *w8 = reinterpret_cast<const int8_t *>(src);
            for (int oc = 0; oc < d.out_c; ++oc) {
                int64_t sum = 0;
                for (int k = 0; k < 27; ++k) sum += w8[oc * 27 + k];
                acc[oc] = (int32_t)(-(int64_t)d.in_zp * sum);
            }
        }
        if (algo == SHL_MI355X_ALGO_DW && p->kstride == 12) {
            const int8_t *w8 = reinterpret_cast<const int8_t *>(src);  // [9][C]
            for (int c = 0; c < d.in_c; ++c) {
                int64_t sum = 0;
                for (int tap = 0; tap < 9; ++tap) sum += w8[(size_t)tap * d.in_c + c];
                acc[c] = (int32_t)(-(int64_t)d.in_zp * sum);
            }
        }
        if (algo == SHL_MI355X_ALGO_IGEMM && d.dtype == SHL_MI355X_I8) {
            // padding is materialised as zp_in, so sum(q*w) carries zp_in*sum(w) for EVERY tap
            const int K = d.kernel_h * d.kernel_w * d.in_c;
            for (int oc = 0; oc < d.out_c; ++oc) {
                const int8_t *row = reinterpret_cast<const int8_t *>(host.data() + p->off_w) +
                                    (size_t)oc * p->kstride;
                int64_t s = 0;
                for (int k = 0; k < K; ++k) s += row[k];
                acc[oc] = (int32_t)(-(int64_t)d.in_zp * s);
            }
        }
    }
    if (algo == SHL_MI355X_ALGO_IGEMM && d.layout == SHL_MI355X_NCHW && d.batch > 0) {
        const size_t in_b = (size_t)d.batch * d.in_c * d.in_h * d.in_w * es;
        const size_t out_b = (size_t)d.batch * d.out_c * d.out_h * d.out_w * es;
        if (hipMalloc((void **)&p->scratch_in, in_b) != hipSuccess ||
            hipMalloc((void **)&p->scratch_out, out_b) != hipSuccess) {
            (void)hipFree(p->scratch_in);
            (void)hipFree(p->block);
            free(p);
            set_error("conv_plan_create: cannot allocate the NCHW re-layout scratch");
            (void)hipGetLastError();
            return SHL_MI355X_ENOMEM;
        }
    }
    e = hipMemcpyAsync(p->block, host.data(), p->block_bytes, hipMemcpyHostToDevice,
                       (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);  // host staging dies here
    if (e != hipSuccess) {
        (void)hipFree(p->block);
        free(p);
        return hip_fail(e, "upload(plan block)");
    }
    const bool rules_need_pix_tab = want_pix_tab;
    if (algo == SHL_MI355X_ALGO_IGEMM && tuning_enabled(d.dtype) && d.batch > 0 && kernel_host && !tune_cached_without_pix_tab(p))
        want_pix_tab = true;  // a candidate of tune_plan
    if (want_pix_tab) {
        const int64_t M = (int64_t)d.batch * d.out_h * d.out_w;
        std::vector<int2> tab((size_t)M);
        const int pix_bytes = d.in_c * es;
        size_t k = 0;
        for (int n = 0; n < d.batch; ++n)
            for (int oy = 0; oy < d.out_h; ++oy)
                for (int ox = 0; ox < d.out_w; ++ox, ++k) {
                    const int y0 = oy * d.stride_h - d.pad_top, x0 = ox * d.stride_w - d.pad_left;
                    uint32_t my = 0, mx = 0;
                    for (int ky = 0; ky < d.kernel_h && ky < 16; ++ky)
                        if ((unsigned)(y0 + ky * d.dilation_h) < (unsigned)d.in_h) my |= 1u << ky;
                    for (int kx = 0; kx < d.kernel_w && kx < 16; ++kx)
                        if ((unsigned)(x0 + kx * d.dilation_w) < (unsigned)d.in_w) mx |= 1u << kx;
                    tab[k].x = (int)((((int64_t)n * d.in_h + y0) * d.in_w + x0) * pix_bytes);
                    tab[k].y = (int)(my | (mx << 16));
                }
        e = M ? hipMalloc((void **)&p->pix_tab, (size_t)M * sizeof(int2)) : hipSuccess;
        if (e == hipSuccess && M) e = hipMemcpyAsync(p->pix_tab, tab.data(), (size_t)M * sizeof(int2), hipMemcpyHostToDevice, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) {
            (void)hipFree(p->pix_tab);
            (void)hipFree(p->scratch_in);
            (void)hipFree(p->scratch_out);
            (void)hipFree(p->block);
            free(p);
            return hip_fail(e, "upload(pixel address table)");
        }
    }
    if (algo == SHL_MI355X_ALGO_IGEMM && kernel_host) tune_plan(p, (hipStream_t)stream, rules_need_pix_tab);
    *plan_out = p;
    return SHL_MI355X_OK;
}

int shl_mi355x_conv_plan_create_dw_channel(const struct shl_mi355x_conv_desc *desc, const void *kernel_host,
                                           const float *kernel_scale, const int32_t *kernel_zp,
                                           const int32_t *bias_i32, float in_scale, float out_scale_ms,
                                           void *stream, shl_mi355x_conv_plan **plan_out)
{
    if (!desc || !plan_out || !kernel_host || !kernel_scale || !kernel_zp) {
        set_error("conv_plan_create_dw_channel: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    *plan_out = nullptr;
    const shl_mi355x_conv_desc &d = *desc;
    int rc = validate(d);
    if (rc != SHL_MI355X_OK || d.layout != SHL_MI355X_NCHW || d.dtype != SHL_MI355X_I8 || d.group != d.in_c ||
        !(out_scale_ms > 0.0f)) {
        set_error("conv_plan_create_dw_channel: int8 NCHW depthwise descriptor with a positive output scale expected");
        return rc != SHL_MI355X_OK ? rc : SHL_MI355X_EINVAL;
    }
    shl_mi355x_conv_plan *p = (shl_mi355x_conv_plan *)calloc(1, sizeof(*p));
    if (!p) return SHL_MI355X_ENOMEM;
    p->desc = d;
    p->algo = SHL_MI355X_ALGO_DW_CHANNEL;
    p->kernel_name = "dwconv_channel_nchw_i8_int64acc";
    p->ch_in_scale = in_scale;
    p->ch_out_scale = out_scale_ms;
    p->ch_has_bias = bias_i32 != nullptr;
    p->inv_out_scale = 1.0f / d.out_scale;
    const size_t w_bytes = (size_t)d.out_c * d.kernel_h * d.kernel_w;
    const size_t tab_bytes = align_up((size_t)d.out_c, 128) * 4;
    p->off_w = 0;
    p->off_acc = align_up(w_bytes, 256);
    p->off_mult = p->off_acc + tab_bytes;
    p->off_bias = p->off_mult + tab_bytes;
    p->off_pad = p->off_bias + tab_bytes;
    p->block_bytes = p->off_pad + PAD_PAGE_BYTES;
    hipError_t e = hipMalloc((void **)&p->block, p->block_bytes);
    if (e != hipSuccess) {
        free(p);
        return hip_fail(e, "hipMalloc(plan block)");
    }
    std::vector<char> host(p->block_bytes, 0);
    memcpy(host.data() + p->off_w, kernel_host, w_bytes);
    memcpy(host.data() + p->off_acc, kernel_zp, (size_t)d.out_c * 4);
    memcpy(host.data() + p->off_mult, kernel_scale, (size_t)d.out_c * 4);
    if (bias_i32) memcpy(host.data() + p->off_bias, bias_i32, (size_t)d.out_c * 4);
    e = hipMemcpyAsync(p->block, host.data(), p->block_bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) {
        (void)hipFree(p->block);
        free(p);
        return hip_fail(e, "upload(plan block)");
    }
    *plan_out = p;
    return SHL_MI355X_OK;
}

int shl_mi355x_conv_plan_destroy(shl_mi355x_conv_plan *plan)
{
    if (!plan) return SHL_MI355X_OK;
    (void)hipFree(plan->scratch_in);
    (void)hipFree(plan->scratch_out);
    (void)hipFree(plan->pix_tab);
    hipError_t e = hipFree(plan->block);
    free(plan);
    if (e != hipSuccess) return hip_fail(e, "hipFree(plan block)");
    return SHL_MI355X_OK;
}

int shl_mi355x_conv_plan_algo(const shl_mi355x_conv_plan *plan) { return plan ? plan->algo : SHL_MI355X_EINVAL; }

const char *shl_mi355x_conv_plan_kernel_name(const shl_mi355x_conv_plan *plan)
{
    return plan ? plan->kernel_name : "";
}

size_t shl_mi355x_conv_plan_bytes(const shl_mi355x_conv_plan *plan)
{
    if (!plan) return 0;
    size_t n = plan->block_bytes;
    if (plan->pix_tab) n += (size_t)plan->desc.batch * plan->desc.out_h * plan->desc.out_w * sizeof(int2);
    if (plan->scratch_in) {
        const shl_mi355x_conv_desc &d = plan->desc;
        const size_t es = d.dtype == SHL_MI355X_I8 ? 1 : 2;
        n += ((size_t)d.batch * d.in_c * d.in_h * d.in_w + (size_t)d.batch * d.out_c * d.out_h * d.out_w) * es;
    }
    return n;
}

void *shl_mi355x_conv_plan_const_block(shl_mi355x_conv_plan *plan, size_t *bytes)
{
    if (!plan) return nullptr;
    if (bytes) *bytes = plan->block_bytes;
    return plan->block;
}

/* after the constant block was overwritten by a broadcast: take the sender's epilogue choices from its flags record */
int shl_mi355x_conv_plan_adopt_block(shl_mi355x_conv_plan *plan, void *stream)
{
    if (!plan) return SHL_MI355X_EINVAL;
    if (!plan->off_flags) return SHL_MI355X_OK;  // plans without a record (depthwise channel ids) carry no host-derived choices
    PlanFlags f;
    hipError_t e = hipMemcpyAsync(&f, plan->block + plan->off_flags, sizeof(f), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "read back the flags record");
    if (f.magic != PLAN_FLAGS_MAGIC) {
        set_error("conv_plan_adopt_block: the block carries no flags record (sender built from another library version?)");
        return SHL_MI355X_EINVAL;
    }
    // the kernel family, the packed row length, the section offsets and the row-patch wave roles decide how the block's
    // bytes are read: a sender that planned otherwise (its tables put it on the direct kernel, another batch another
    // geometry) cannot be adopted -- its epilogue flags belong to ITS kernel (div_exact = div_fma = 0 is only legal there)
    if (f.algo != plan->algo || f.kstride != plan->kstride || f.block_bytes != plan->block_bytes || f.pt_geom != plan->pt_geom) {
        set_error("conv_plan_adopt_block: sender planned algo %d / K row %d / %llu bytes / patch roles 0x%x, receiver algo %d / %d / "
                  "%llu / 0x%x -- create the receiver's plan from the same descriptor, batch and table regime",
                  f.algo, f.kstride, (unsigned long long)f.block_bytes, (unsigned)f.pt_geom, plan->algo, plan->kstride,
                  (unsigned long long)plan->block_bytes, (unsigned)plan->pt_geom);
        return SHL_MI355X_EINVAL;
    }
    plan->div_exact = f.div_exact, plan->div_fma = f.div_fma, plan->act_clamp = f.act_clamp;
    plan->clamp_lo = f.clamp_lo, plan->clamp_hi = f.clamp_hi, plan->inv_out_scale = f.inv_out_scale;
    return SHL_MI355X_OK;
}

}  // extern "C"

// kernel argument block of one forward pass of `plan`
static int fill_args(const shl_mi355x_conv_plan *plan, const void *input_dev, void *output_dev, int32_t batch,
                     shl::ConvArgs &a)
{
    using namespace shl;
    const shl_mi355x_conv_desc &d = plan->desc;
    memset(&a, 0, sizeof(a));
    a.in = input_dev;
    a.out = output_dev;
    a.w = plan->block + plan->off_w;
    a.w_frag = plan->off_wfrag ? plan->block + plan->off_wfrag : nullptr;
    a.w_patch = plan->off_wpatch ? plan->block + plan->off_wpatch : nullptr;
    a.pt_geom = plan->pt_geom;
    a.acc_init = reinterpret_cast<const int32_t *>(plan->block + plan->off_acc);
    a.mult = reinterpret_cast<const float *>(plan->block + plan->off_mult);
    a.bias = reinterpret_cast<const float *>(plan->block + plan->off_bias);
    a.N = batch > 0 ? batch : d.batch;
    a.H = d.in_h; a.W = d.in_w; a.C = d.in_c;
    a.Ho = d.out_h; a.Wo = d.out_w; a.Co = d.out_c;
    a.Kh = d.kernel_h; a.Kw = d.kernel_w;
    a.sh = d.stride_h; a.sw = d.stride_w;
    a.pt = d.pad_top; a.pl = d.pad_left;
    a.dh = d.dilation_h; a.dw = d.dilation_w;
    a.group = d.group;
    const int64_t M = (int64_t)a.N * a.Ho * a.Wo;
    if (M > 0x7FFFFFFF) {
        set_error("conv_forward: N*Ho*Wo exceeds 2^31-1");
        return SHL_MI355X_EINVAL;
    }
    a.M = (int32_t)M;
    a.kchunks = plan->kchunks;
    a.kstride = plan->kstride;
    a.cchunks = plan->cchunks;
    a.in_zp = d.in_zp;
    a.act = d.act;
    a.out_scale = d.out_scale;
    a.out_zp_f = (float)d.out_zp;
    a.out_zp = d.out_zp;
    const float os = d.out_scale;
    a.scale_out = (d.dtype == SHL_MI355X_F16) && (os - 1.0f > 1.1920929e-07f || 1.0f - os > 1.1920929e-07f);
    a.inv_out_scale = plan->inv_out_scale;
    a.div_exact = plan->div_exact;
    a.div_fma = plan->div_fma;
    a.act_clamp = plan->act_clamp;
    a.clamp_lo = plan->clamp_lo;
    a.clamp_hi = plan->clamp_hi;
    a.pad_page = plan->block + plan->off_pad;
    // the per-pixel table covers desc.batch images: a larger batch runs on the kernels that do their own index arithmetic
    a.pix_tab = a.N <= d.batch ? plan->pix_tab : nullptr;
    a.ch_in_scale = plan->ch_in_scale;
    a.ch_out_scale = plan->ch_out_scale;
    a.ch_has_bias = plan->ch_has_bias;
    {
        static const char *dbg = getenv("SHL_MI355X_DEBUG");
        a.debug = dbg ? atoi(dbg) : 0;
    }
    return SHL_MI355X_OK;
}

extern "C" {

static int conv_forward_impl(const shl_mi355x_conv_plan *plan, const void *input_dev, void *output_dev, int32_t batch,
                             void *stream);

int shl_mi355x_conv_forward(const shl_mi355x_conv_plan *plan, const void *input_dev,
                            void *output_dev, int32_t batch, void *stream)
{
    if (!plan || !input_dev || !output_dev) {
        set_error("conv_forward: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    // the family measured best for this plan's own batch (other batches: the selection rules)
    const bool tuned = plan->variant && (batch <= 0 || batch == plan->desc.batch);
    if (tuned) igemm_set_plan_variant(plan->variant);
    const int rc = conv_forward_impl(plan, input_dev, output_dev, batch, stream);
    if (tuned) igemm_set_plan_variant(nullptr);
    return rc;
}

static int conv_forward_impl(const shl_mi355x_conv_plan *plan, const void *input_dev, void *output_dev, int32_t batch,
                             void *stream)
{
    const shl_mi355x_conv_desc &d = plan->desc;
    ConvArgs a;
    int frc = fill_args(plan, input_dev, output_dev, batch, a);
    if (frc != SHL_MI355X_OK) return frc;
    if (a.M == 0) return SHL_MI355X_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (plan->algo) {
        case SHL_MI355X_ALGO_IGEMM: {
            if (d.layout == SHL_MI355X_NHWC) return launch_conv_igemm(a, d.dtype, d.layout, s);
            // a 1x1 spatial extent on both sides (classifier on the pooled map): [N,C,1,1] and
            // [N,1,1,C] are the same bytes, the NHWC kernels apply as they are
            if (a.H * a.W == 1 && a.Ho * a.Wo == 1) return launch_conv_igemm(a, d.dtype, SHL_MI355X_NHWC, s);
            // latency-bound pointwise layers read and write NCHW directly (nchw_small.hip)
            if (!strcmp(igemm_variant(a.M, a.Co), "wave") && conv1x1_nchw_eligible(a)) {
                igemm_note_family("nchw1x1");
                return launch_conv1x1_nchw(a, d.dtype, s);
            }
            // 3x3 "same": the row-patch kernel reads and writes NCHW itself (conv_igemm_patch.hip; int8, and binary16 at stride 1)
            if (a.w_patch) {
                ConvArgs t = a;
                t.in_nchw = t.out_nchw = 1;
                if (!strcmp(igemm_pick_name(t, d.dtype == SHL_MI355X_I8 ? 1 : 2), "patch")) {
                    igemm_note_family("patch");
                    return launch_conv_igemm_patch(t, s);
                }
            }
            // NCHW: [C][HW] -> [HW][C] scratch, NHWC kernel, [HoWo][Co] -> [Co][HoWo]
            if (a.N > d.batch || !plan->scratch_in) {
                set_error("conv_forward: NCHW plan was created for batch %d, got %d", d.batch, a.N);
                return SHL_MI355X_EINVAL;
            }
            const int es = d.dtype == SHL_MI355X_I8 ? 1 : 2;
            int rc = launch_transpose(input_dev, plan->scratch_in, a.N, a.C, a.H * a.W, es, s, 1);
            if (rc != SHL_MI355X_OK) return rc;
            a.in = plan->scratch_in;
            // the tile kernel's epilogue stores NCHW itself; planes whose byte size is not a multiple of
            // 4 (7x7 int8) would fall to element stores there and are cheaper through the re-layout pass
            // (binary16 3x3: when the row-patch kernel takes the NHWC view it is worth the second re-layout pass)
            const bool f16_patch = es == 2 && a.w_patch && !strcmp(igemm_pick_name(a, es), "patch");
            if (!f16_patch && igemm_fuses_nchw_out(a, es) && ((a.Ho * a.Wo * es) & 3) == 0) {
                a.out_nchw = 1;
                return launch_conv_igemm(a, d.dtype, SHL_MI355X_NHWC, s);
            }
            a.out = plan->scratch_out;
            rc = launch_conv_igemm(a, d.dtype, SHL_MI355X_NHWC, s);
            if (rc != SHL_MI355X_OK) return rc;
            return launch_transpose(plan->scratch_out, output_dev, a.N, a.Ho * a.Wo, a.Co, es, s, 0);
        }
        case SHL_MI355X_ALGO_DW:
            if (d.layout == SHL_MI355X_NCHW) return launch_dwconv_nchw(a, d.dtype, s);
            return launch_dwconv(a, d.dtype, d.layout, s);
        case SHL_MI355X_ALGO_STEM:
            return launch_conv_stem(a, s);
        case SHL_MI355X_ALGO_DW_CHANNEL:
            return launch_dwconv_channel(a, s);
        case SHL_MI355X_ALGO_GROUP:
            return launch_conv_group_direct(a, d.dtype, d.layout, s);
        default:
            return launch_conv_direct(a, d.dtype, d.layout,
                                      is_depthwise(d) && d.layout == SHL_MI355X_NHWC, s);
    }
}

int shl_mi355x_debug_trace(uint64_t *host, int32_t count)
{
    if (!host || count <= 0) return SHL_MI355X_EINVAL;
    const char *v = getenv("SHL_MI355X_IGEMM");  // which kernel's stamps: the producer / consumer kernel when it is forced
    if (v && !strcmp(v, "patch")) return patch_read_trace(reinterpret_cast<unsigned long long *>(host), count);
    if (v && !strcmp(v, "pc")) return pc_read_trace(reinterpret_cast<unsigned long long *>(host), count);
    return pp_read_trace(reinterpret_cast<unsigned long long *>(host), count);
}

/* which fused kernel runs the pair (first layer `pw`, second `dw`; 6 is the pair in the other order): 0 none, 1 latency form (pwdw_fused.hip: small grids), 3 stem + depthwise
 * (stemdw_fused.hip), 4 binary16 NCHW (pwdw_f16_nchw.hip), 5 binary16 NCHW stem + depthwise (stemdw_f16_nchw.hip).  (2 was the int8 bandwidth form for large batches: it only
 * broke even with the two stand-alone kernels, attic/README.md.)  6 = depthwise -> pointwise in bandwidth form (dwpw_stream.hip): the
 * 32 / 64 / 128 / 256-channel blocks at large batches. */
static int pwdw_kernel_for(const shl_mi355x_conv_plan *pw, const shl_mi355x_conv_plan *dw, const ConvArgs &a,
                           const ConvArgs &b)
{
    // the other order (6): `pw` is the depthwise layer and `dw` the pointwise one consuming it (dwpw_stream.hip, large batches)
    if (pw->algo == SHL_MI355X_ALGO_DW) {
        if (pw->desc.dtype != SHL_MI355X_I8) return 0;
        // 7: the deep blocks (512 channels; 256 @28) at throughput batches: resident pointwise weights behind a depthwise stage
        // (dwpw_resident.hip) -- asked first: where its rule takes a 256-channel pair the streaming form would re-fetch the
        // pointwise weights per 64 pixels
        if (dwpw_resident_fusable(a, b, pw->kstride == 12, dw->algo == SHL_MI355X_ALGO_IGEMM)) return 7;
        return dwpw_stream_fusable(a, b, pw->kstride == 12, dw->algo == SHL_MI355X_ALGO_IGEMM) ? 6 : 0;
    }
    const int pw_igemm = pw->algo == SHL_MI355X_ALGO_IGEMM, dw_dot4 = dw->algo == SHL_MI355X_ALGO_DW && dw->kstride == 12;
    if (pw->desc.dtype == SHL_MI355X_F16) {  // binary16 NCHW (pwdw_f16_nchw.hip, stemdw_f16_nchw.hip)
        if (dw->algo != SHL_MI355X_ALGO_DW) return 0;
        if (pw_igemm) return pwdw_f16_nchw_fusable(a, b) ? 4 : 0;
        return pw->algo == SHL_MI355X_ALGO_DIRECT && stemdw_f16_nchw_fusable(a, b) ? 5 : 0;
    }
    // int8 NHWC latency forms: at throughput sizes the depthwise layer belongs to the pair with the pointwise layer BEHIND it
    // (dwpw_stream.hip); a chain pairs up one way round
    {
        static const char *sel = getenv("SHL_MI355X_PWDW");  // "2": latency forms without size rules (A/B)
        // (... unless the graph's owner said that this depthwise layer has no such consumer -- MobileNetV2's dw 32 -> pw 16, two
        // consumers, a non-1x1 layer behind it: shl_mi355x_conv_plan_set_no_stream_consumer; without the hint both fusions were lost)
        if (!(sel && sel[0] == '2') && dw_dot4 && (dwpw_stream_takes(b) || dwpw_resident_takes(b)) && !dw->no_stream_consumer) return 0;
    }
    if (pw->algo == SHL_MI355X_ALGO_STEM) return dw_dot4 && stemdw_fusable(a, b) ? 3 : 0;  // stem + depthwise
    return pwdw_fusable(a, b, pw_igemm, dw_dot4) ? 1 : 0;
}

int shl_mi355x_conv_plan_set_no_stream_consumer(shl_mi355x_conv_plan *plan, int32_t on)
{
    if (!plan) return SHL_MI355X_EINVAL;
    plan->no_stream_consumer = on ? 1 : 0;
    return SHL_MI355X_OK;
}

/* global_avgpool2d + the convolution / fullyconnected layer on the pooled map in one launch (conv_gemv.hip:pool_gemv_i8_kernel) */
int shl_mi355x_pool_conv_fusable(const shl_mi355x_conv_plan *plan, int32_t batch, int32_t pixels)
{
    if (!plan || plan->algo != SHL_MI355X_ALGO_IGEMM || plan->desc.dtype != SHL_MI355X_I8) return 0;
    static const char *off = getenv("SHL_MI355X_NO_FUSION");
    if (off && off[0] == '1') return 0;
    // OFF unless SHL_MI355X_POOLGEMV=1 (read per call).  Measured in round 6 (tools/dev/pool_gemv_time.py, session_ab.py;
    // profiles/r06_notes.md): 7 x 7 x 1024 -> 1000: avgpool 4.19 us + GEMV 2.45 us = 6.77 us as two launches, 8.52 us fused
    // (every workgroup that needs the pooled vector repeats the pooling: 16 x the additions per CU), csinn_session_run
    // 84.5 -> 87.0 us.  Kept as a tested, bit-exact entry point; the session does not take it by default.
    const char *sel = getenv("SHL_MI355X_POOLGEMV");
    if (!(sel && sel[0] == '1')) return 0;
    if (plan->desc.in_h * plan->desc.in_w != 1 || igemm_env_override()) return 0;
    ConvArgs a;
    static char dummy[16];
    if (fill_args(plan, dummy, dummy, batch, a) != SHL_MI355X_OK) return 0;
    return pool_gemv_pick(a, pixels) ? 1 : 0;
}

int shl_mi355x_pool_conv_forward(const shl_mi355x_conv_plan *plan, const void *input_dev, void *output_dev, int32_t batch,
                                 int32_t pixels, float in_scale, int32_t in_zp, float mid_scale, int32_t mid_zp, void *stream)
{
    if (!plan || !input_dev || !output_dev) {
        set_error("pool_conv_forward: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    if (!shl_mi355x_pool_conv_fusable(plan, batch, pixels) || mid_zp != plan->desc.in_zp) {
        set_error("pool_conv_forward: the pair does not qualify for the fused kernel");
        return SHL_MI355X_ENOTSUP;
    }
    ConvArgs a;
    int rc = fill_args(plan, input_dev, output_dev, batch, a);
    if (rc != SHL_MI355X_OK) return rc;
    if (a.M == 0) return SHL_MI355X_OK;
    return launch_pool_gemv(a, pixels, in_scale, in_zp, mid_scale, mid_zp, (hipStream_t)stream);
}

/* pointwise 1x1 + the global_avgpool2d consuming it, fused into one launch (conv1x1_latency.hip) */
int shl_mi355x_conv_pool_fusable(const shl_mi355x_conv_plan *plan, int32_t batch)
{
    using namespace shl;
    if (!plan || batch <= 0) return 0;
    const char *sel = getenv("SHL_MI355X_CONVPOOL");  // "0": keep the two launches (read per call: tests, A/B)
    if (sel && sel[0] == '0') return 0;
    if (plan->desc.dtype != SHL_MI355X_I8 || plan->desc.layout != SHL_MI355X_NHWC || plan->algo != SHL_MI355X_ALGO_IGEMM) return 0;
    if (igemm_env_override()) return 0;
    ConvArgs a;
    static char dummy[16];
    if (fill_args(plan, dummy, dummy, batch, a) != SHL_MI355X_OK) return 0;
    return conv1x1_pool_fusable(a) ? 1 : 0;
}

int shl_mi355x_conv_pool_forward(const shl_mi355x_conv_plan *plan, const void *input_dev, void *map_dev, void *pool_dev, int32_t batch,
                                 float mid_scale, int32_t mid_zp, float out_scale, int32_t out_zp, void *stream)
{
    using namespace shl;
    if (!plan || !input_dev || !pool_dev) {
        set_error("conv_pool_forward: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    if (!shl_mi355x_conv_pool_fusable(plan, batch)) {
        set_error("conv_pool_forward: the pair does not qualify for the fused kernel");
        return SHL_MI355X_ENOTSUP;
    }
    ConvArgs a;
    static char dummy[16];
    int rc = fill_args(plan, input_dev, map_dev ? map_dev : dummy, batch, a);
    if (rc != SHL_MI355X_OK) return rc;
    if (a.M == 0) return SHL_MI355X_OK;
    return launch_conv1x1_pool(a, pool_dev, mid_scale, (float)mid_zp, out_scale, (float)out_zp, map_dev ? 1 : 0, (hipStream_t)stream);
}

/* pointwise 1x1 + the depthwise 3x3 consuming it, fused into one launch */
int shl_mi355x_pwdw_form(const shl_mi355x_conv_plan *pw, const shl_mi355x_conv_plan *dw, int32_t batch)
{
    if (!dw || !pw) return 0;
    static const char *off = getenv("SHL_MI355X_NO_FUSION");
    static const char *sel = getenv("SHL_MI355X_PWDW");  // "0": keep pointwise and depthwise launches apart
    if ((off && off[0] == '1') || (sel && sel[0] == '0')) return 0;
    if (dw->desc.dtype != pw->desc.dtype || dw->desc.layout != pw->desc.layout) return 0;
    // int8 NHWC (pwdw_fused.hip, stemdw_fused.hip) or binary16 NCHW (pwdw_f16_nchw.hip)
    const bool i8_nhwc = pw->desc.dtype == SHL_MI355X_I8 && pw->desc.layout == SHL_MI355X_NHWC;
    const bool f16_nchw = pw->desc.dtype == SHL_MI355X_F16 && pw->desc.layout == SHL_MI355X_NCHW;
    if (!i8_nhwc && !f16_nchw) return 0;
    ConvArgs a, b;
    static char dummy[16];
    if (fill_args(pw, dummy, dummy, batch, a) != SHL_MI355X_OK || fill_args(dw, dummy, dummy, batch, b) != SHL_MI355X_OK)
        return 0;
    return pwdw_kernel_for(pw, dw, a, b);
}

int shl_mi355x_pwdw_fusable(const shl_mi355x_conv_plan *pw, const shl_mi355x_conv_plan *dw, int32_t batch)
{
    return shl_mi355x_pwdw_form(pw, dw, batch) ? 1 : 0;
}

int shl_mi355x_pwdw_forward(const shl_mi355x_conv_plan *pw, const shl_mi355x_conv_plan *dw, const void *input_dev,
                            void *output_dev, int32_t batch, void *stream)
{
    if (!dw || !pw || !input_dev || !output_dev) {
        set_error("pwdw_forward: NULL argument");
        return SHL_MI355X_EINVAL;
    }
    if (!shl_mi355x_pwdw_fusable(pw, dw, batch)) {
        set_error("pwdw_forward: the pair does not qualify for the fused kernel");
        return SHL_MI355X_ENOTSUP;
    }
    ConvArgs a, b;
    int rc = fill_args(pw, input_dev, output_dev, batch, a);
    if (rc == SHL_MI355X_OK) rc = fill_args(dw, input_dev, output_dev, batch, b);
    if (rc != SHL_MI355X_OK) return rc;
    if (b.M == 0) return SHL_MI355X_OK;
    switch (pwdw_kernel_for(pw, dw, a, b)) {
        case 7: return launch_dwpw_resident(a, b, (hipStream_t)stream);
        case 6: return launch_dwpw_stream(a, b, (hipStream_t)stream);
        case 5: return launch_stemdw_f16_nchw(a, b, (hipStream_t)stream);
        case 4: return launch_pwdw_f16_nchw(a, b, (hipStream_t)stream);
        case 3: return launch_stemdw_fused(a, b, (hipStream_t)stream);
        default: return launch_pwdw_fused(a, b, (hipStream_t)stream);
    }
}

}  // extern "C"
