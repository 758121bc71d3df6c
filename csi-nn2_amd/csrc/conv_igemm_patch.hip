// conv_igemm_patch.hip -- 3x3 stride-1 "same" int8 convolution as an implicit GEMM whose im2col matrix exists
// only as ONE staged row patch in LDS, read NHWC or NCHW natively (row a5 of SURVEY 8: no re-layout pass).
//
// Why another kernel (profiles/r02_notes.md): the block-tile kernels fetch every input byte nine times through
// the CU's L2 -> LDS path (one piece per filter tap), and that path -- ~29 B per cycle next to busy matrix
// cores -- paced their K loops at half the MFMA rate; ResNet-50's layers at batch 128 are ONE tile per CU, so
// nothing overlapped the fixed costs either.  Here
//   * a workgroup owns R whole output rows (R * W <= 416 pixels = 13 MFMA pixel blocks) of TN = 32 * OB output
//     channels.  Per 64 / 128-channel stage the rows' input patch (+ one halo row above / below, one shared
//     padding column per row, padding rows between images) is staged ONCE as [patch pixel][KC + 16 bytes]: a
//     filter tap is a constant LDS address offset (ty * (W + 1) + tx) * pitch, so all nine taps -- the whole
//     im2col expansion -- are served from LDS and the L2 -> LDS traffic drops ~6x.  The 16-byte pad slot per
//     pixel makes the pitch an odd number of 16-byte slots: every ds_read_b128 lane group hits 64 distinct
//     banks whatever the tap shift (no XOR swizzle, which would not commute with the shift);
//   * weights never touch LDS: a wave owns ONE 32-channel block (x a K part), nobody else needs its A
//     fragments, so they stream global -> VGPR from a plan-time copy in fragment order (one coalesced 1-KiB
//     load per K step, three steps of look-ahead);
//   * four waves, one per SIMD, up to 512 registers each: 13 x 16 accumulators, an 8-deep ring of B
//     fragments read 7 MFMAs ahead, and the staging of the NEXT stage (global -> VGPR -> LDS) in the same
//     instruction stream under the MFMAs -- NHWC: 16-byte pieces as they are; NCHW: a lane loads 16 pixels x
//     16 channels of plane runs (unaligned 16-byte loads: tools/probes/unaligned.hip), transposes 16 x 16
//     bytes in registers (128 v_perm_b32 per 256 bytes, ~3 % of a stage's issue slots) and writes pixel-major.
//     With one staging per nine taps this costs less than one LDS-DMA piece per tap did;
//   * wave roles (PG pixel groups x OB channel blocks x KP K parts = 4) are chosen so that a layer is ~one
//     round of tiles on the 256 CUs: ResNet-50 at batch 128 = exactly 256 (or 512) tiles for every 3x3 layer
//     (64ch @56: 2 x 2 x 1, 128 @28: 1 x 4 x 1, 256 @14: 1 x 2 x 2, 512 @7: 1 x 1 x 4).  K parts are summed
//     exactly (int32) through LDS in the epilogue;
//   * epilogue from registers: a wave has ONE channel block, so its per-channel tables are 2 (NCHW: lane =
//     channel) or 32 (NHWC) registers; v_permlane32_swap gives every lane 16 consecutive bytes of the output
//     tensor's fastest dimension -> one 16-byte store per MFMA block per lane in either layout.
// Numerical contract: common.h (exact int32 sums, two-rounding fp32 epilogue) -- bit-identical to every other
// int8 kernel of the library and to oracle formulation X.
//
// Replaces shl_ref_conv2d_nhwc_f32 / shl_ref_conv2d_nchw_f32 + conv_im2col_sgemm_avx
// (source/reference/convolution.c:28-139, conv_avx.h:109-1008) inside shl_ref_conv2d_quant (:370-400).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "conv_igemm_patch_kernel.h"

namespace shl {

// =================================================================================================== host side
static int pt_esize(const shl_mi355x_conv_desc &d) { return d.dtype == SHL_MI355X_I8 ? 1 : 2; }

bool patch_supports(const shl_mi355x_conv_desc &d)
{
    if (d.group != 1) return false;
    if (d.dtype == SHL_MI355X_F16) {
        // binary16: NCHW layers natively (8 x 8 transposition of two-byte elements in the staging; stride 1 and the stride-2 form), or
        // -- stride 1, when the NCHW item budget does not cover a stage -- on the NHWC view of conv_forward's re-layout path
        static const char *f16_env = getenv("SHL_MI355X_PATCH_F16");  // "0": off (A/B)
        if (f16_env && f16_env[0] == '0') return false;
        if ((d.stride_h != 1 || d.stride_w != 1) && d.layout != SHL_MI355X_NCHW) return false;
    } else if (d.dtype != SHL_MI355X_I8) {
        return false;
    }
    if (d.kernel_h != 3 || d.kernel_w != 3 || d.dilation_h != 1 || d.dilation_w != 1) return false;
    if (d.pad_top != 1 || d.pad_left != 1) return false;
    if (d.stride_h == 2 && d.stride_w == 2) {
        // the stride-2 form: even planes, so that no tap reaches the bottom / right padding
        if (d.in_h % 2 || d.in_w % 2 || d.out_h != d.in_h / 2 || d.out_w != d.in_w / 2) return false;
    } else if (d.stride_h != 1 || d.stride_w != 1 || d.out_h != d.in_h || d.out_w != d.in_w) {
        return false;
    }
    if ((d.in_c * pt_esize(d)) % 64 != 0 || d.in_w > PT_PIX) return false;
    if (d.layout == SHL_MI355X_NHWC && d.out_c % 16 != 0) return false;  // 16-byte stores of 16 (binary16: 2 x 8) channels (NCHW: lane = channel)
    if (d.dtype == SHL_MI355X_I8 && (d.in_zp < -128 || d.in_zp > 127)) return false;
    return true;
}

namespace {

struct PatchShape {
    int rows, prows, bufb, lds, nt_m, nt_n, nitc, spr;
    int pair_dn;  // pair mode: the second tile of a workgroup lies this many images further on (0 = off)
};

// geometry of a forward pass with batch n; false when the patch does not fit LDS / the staging item budget
// shape for a given number of rows per pixel group: 1 fits, 0 too large for LDS / the staging item budget (try fewer
// rows), -1 the NCHW item rounds do not fit (eight waves: retry with four)
int patch_shape_rows(int n, int H, int W, int C, int Co, bool nchw, int geom, int rows, PatchShape *ps)
{
    const int kc = PT_KC(geom), pg = PT_PG(geom), ob = PT_OB(geom), kp = PT_KP(geom);
    const int pitch = kc + 16, slots = kc / 16;
    const int tr = rows * pg;
    // stride 2 = the stride-1 geometry on the half-resolution grid (conv_igemm_patch_kernel.h: kS2): H, W below are the grid's,
    // TW the tensor's row length (NCHW runs are rows of the tensor)
    const bool s2 = PT_S2(geom) != 0;
    const int TW = W;
    if (s2) H /= 2, W /= 2;
    const int kcp = s2 ? kc / 4 : kc;  // channels of the tensor per stage
    const int64_t total_rows = (int64_t)n * H;
    const int nt_m = (int)((total_rows + tr - 1) / tr);
    int prows = 0, runs = 0;
    for (int tm = 0; tm < nt_m; ++tm) {  // exact capacity over the tiles of this batch
        const int64_t r0 = (int64_t)tm * tr, rl = (r0 + tr < total_rows ? r0 + tr : total_rows) - 1;
        const int64_t v0 = r0 + r0 / H - 1;
        const int p = (int)(rl + rl / H - v0 + 2);
        prows = p > prows ? p : prows;
        const int64_t nf = (v0 + 1) / (H + 1), nl = (v0 + p - 1) / (H + 1);  // first / last image with rows in the patch
        const int ru = (int)(nl - nf + 1);
        runs = ru > runs ? ru : runs;
    }
    const int px = prows * (W + 1) + 1;
    const int bufb = (px * pitch + 255) & ~255;
    int lds = 2 * bufb + PT_TRASH + PT_TABLES;
    if (prows > 512 || tr > 512 || runs > 250) return 0;  // row tables of the prologue
    if (kp > 1 && lds < 4 * 8 * 4096) lds = 4 * 8 * 4096;
    if (lds > PT_LDS_MAX) return 0;
    if (!nchw && (int64_t)prows * W * slots > PT_NIT * 256) return 0;
    int nitc = 1, spr = 1;
    if (nchw) {
        const int maxrun = (rows * pg + 2 < H ? rows * pg + 2 : H) * (s2 ? 2 * TW : W);
        const int es = PT_F16(geom) ? 2 : 1;  // a 16-byte segment of a plane run is 16 / es pixels, an item 8 (16) channels of it
        spr = (maxrun * es + 15) / 16;
        const int nw8 = PT_NW8(geom);
        const int items = runs * spr * (kcp / ((nw8 ? 8 : 16) * es));
        nitc = (items + (nw8 ? 511 : 255)) / (nw8 ? 512 : 256);
        if (nitc > (nw8 ? 1 : 2)) return -1;
    }
    ps->rows = rows, ps->prows = prows, ps->bufb = bufb, ps->lds = lds, ps->nt_m = nt_m;
    ps->nt_n = (((Co + 31) / 32) + ob - 1) / ob;
    ps->nitc = nitc, ps->spr = spr;
    // pair mode (eight waves, one K part): a single-stage layer whose tiles need more than one round on the 256 CUs;
    // tile t and tile t + nt_m / 2 must be congruent (the same rows of another image) and no tile ragged.  One
    // prologue for two tiles = ~1.7 tile times per workgroup
    ps->pair_dn = 0;
    static const char *pair_env = getenv("SHL_MI355X_PATCH_PAIR");  // "0": off (A/B), "1": whenever the tiles pair up (tests)
    const int64_t tiles = (int64_t)nt_m * ps->nt_n;
    const bool pays = (pair_env && pair_env[0] == '1') || (double)((tiles + 511) / 512) * 1.7 < (double)((tiles + 255) / 256);
    // (13-block tiles only: the small-tile instantiations have no pair form -- launched on half the tiles they would leave the
    // other half of the output unwritten; round 5: reachable through SHL_MI355X_PATCH=1,4,1,7 on single-stage layers)
    if (!(pair_env && pair_env[0] == '0') && !PT_F16(geom) && !s2 && PT_NW8(geom) && PT_NBT(geom) == PT_NB && C == kc && kp == 1 && nt_m % 2 == 0 && total_rows % tr == 0 &&
        ((int64_t)(nt_m / 2) * tr) % H == 0 && pays)
        ps->pair_dn = (int)((int64_t)(nt_m / 2) * tr / H);
    return 1;
}

bool patch_shape_uncached(int n, int H, int W, int C, int Co, bool nchw, int geom, PatchShape *ps);

// geometry of a forward pass with batch n; false when the patch does not fit LDS / the staging item budget.  The
// search walks the tiles of every candidate row count: layer-mode callers launch per call, so the last few answers
// are kept (per thread)
bool patch_shape(int n, int H, int W, int C, int Co, bool nchw, int geom, PatchShape *ps)
{
    struct Entry {
        int key[7];
        bool ok;
        PatchShape ps;
    };
    geom &= ~(1 << 22);  // PT_NT is a launch flag, not geometry
    static thread_local Entry cache[16];
    static thread_local int used = 0, next = 0;
    const int key[7] = {n, H, W, C, Co, nchw ? 1 : 0, geom};
    for (int i = 0; i < used; ++i)
        if (!memcmp(cache[i].key, key, sizeof(key))) {
            *ps = cache[i].ps;
            return cache[i].ok;
        }
    Entry &e = cache[next];
    next = (next + 1) % 16;
    used = used < 16 ? used + 1 : 16;
    memcpy(e.key, key, sizeof(key));
    e.ps = PatchShape{};
    e.ok = patch_shape_uncached(n, H, W, C, Co, nchw, geom, &e.ps);
    *ps = e.ps;
    return e.ok;
}

bool patch_shape_uncached(int n, int H, int W, int C, int Co, bool nchw, int geom, PatchShape *ps)
{
    const int kc = PT_KC(geom), pg = PT_PG(geom);
    const bool s2 = PT_S2(geom);
    if (!kc || C % (s2 ? kc / 4 : kc) != 0) return false;
    const int HO = s2 ? H / 2 : H, WO = s2 ? W / 2 : W;  // output plane
    const int64_t total_rows = (int64_t)n * HO;
    if ((int64_t)n * H + n >= (1 << 22) || (int64_t)n * H * W >= (1 << 24) || (int64_t)n * C >= (1 << 24)) return false;  // 24-bit multiplies
    if ((int64_t)n * H * W * C >= (1ll << 31) - 65536) return false;  // 32-bit source offsets
    int rows = PT_NBT(geom) * 32 / WO;  // a pixel group fits its wave role's blocks
    if (rows < 1) return false;
    if ((int64_t)rows * pg > total_rows) rows = (int)((total_rows + pg - 1) / pg);
    int top = 0;  // the most rows that fit
    for (; rows >= 1 && !top; --rows) {
        const int rc = patch_shape_rows(n, H, W, C, Co, nchw, geom, rows, ps);
        if (rc == 1) top = rows;
    }
    if (!top) return false;
    // prefer tiles that start on image boundaries (whole images per tile, or a whole number of tiles per image) when
    // that costs no extra round of tiles on the 256 CUs: fewer padding rows and NCHW runs (512 -> 512 @7: 56 rows = 8
    // images instead of 57)
    PatchShape best = *ps;
    const int64_t rounds_top = ((int64_t)best.nt_m * best.nt_n + 255) / 256;
    for (int r = top; r >= 1 && 8 * r >= 7 * top; --r) {
        const int tr = r * pg;
        if (tr % HO != 0 && HO % tr != 0) continue;
        PatchShape t;
        if (patch_shape_rows(n, H, W, C, Co, nchw, geom, r, &t) != 1) continue;
        if (((int64_t)t.nt_m * t.nt_n + 255) / 256 <= rounds_top) best = t;
        break;
    }
    *ps = best;
    return true;
}

int make_geom(int kc, int pg, int ob, int kp, int nbt = PT_NB)
{
    static const char *env = getenv("SHL_MI355X_PATCH_WAVES");  // "4": one wave per SIMD (A/B, tests)
    const int nw8 = !(env && env[0] == '4');
    return kc | pg << 8 | ob << 12 | kp << 16 | nw8 << 20 | (nbt == 7 ? 1 : nbt == 4 ? 2 : 0) << 24;
}

// small tiles (7 / 4 pixel blocks per wave role): the instantiations that exist (conv_igemm_patch_kernel.h:patch_launch_one)
bool small_tiles_ok(bool nchw, bool f16, int kc, int pg, int ob, int kp, int g)
{
    return nchw && kc == 128 && pg == 1 && ob == 4 && kp == 1 && PT_NW8(g);
}

}  // namespace

// wave roles for a batch: the candidate with the fewest (rounds of tiles on 256 CUs) x (K share), ties to fewer
// K parts (no exchange) and more channel blocks per tile (fewer staged patches)
int patch_choose_geom(const shl_mi355x_conv_desc &d, int32_t batch)
{
    if (!patch_supports(d) || batch <= 0) return 0;
    static const char *env = getenv("SHL_MI355X_PATCH");  // "pg,ob,kp" forces the roles (tests, A/B)
    const bool s2 = d.stride_h == 2;
    const bool f16 = d.dtype == SHL_MI355X_F16;
    const int cbytes = d.in_c * pt_esize(d);  // the kernel's "channels" are the bytes of a pixel
    int kc = (cbytes % 128 == 0 || s2) ? 128 : 64;  // stride 2: a patch pixel is four planes of 32 channels
    int u = kc / 32;
    bool nchw = d.layout == SHL_MI355X_NCHW;  // (binary16 NCHW layers whose stages no NCHW geometry covers: the NHWC view, below)
    // NHWC keeps its stride-2 layers on the block-tile kernels (28 / 22 / 23 us for ResNet-50's at batch 128); NCHW saves the
    // re-layout passes around them
    if (s2 && !nchw) return 0;
    if (env && strchr(env, ',')) {  // ("0" / "1" alone are the on / off switch; round 5: "1,4,1" used to be read as "1")
        int pg = 0, ob = 0, kp = 0, nbt = PT_NB;
        if (sscanf(env, "%d,%d,%d,%d", &pg, &ob, &kp, &nbt) >= 3 && pg * ob * kp == 4 && u % kp == 0 && pg != 4 && !(pg == 2 && kp == 2) && !(s2 && kp != 1)) {
            PatchShape ps;
            if (nbt != 7 && nbt != 4) nbt = PT_NB;
            if (nbt != PT_NB && !small_tiles_ok(nchw, f16, kc, pg, ob, kp, make_geom(kc, pg, ob, kp))) nbt = PT_NB;
            int g = make_geom(kc, pg, ob, kp, nbt) | (s2 ? 1 << 21 : 0) | (f16 ? 1 << 23 | 1 << 20 : 0);
            if (s2 && !PT_NW8(g)) return 0;  // the stride-2 form has eight waves
            if (!s2 && !f16 && !patch_shape(batch, d.in_h, d.in_w, cbytes, d.out_c, nchw, g, &ps)) g &= ~(1 << 20);  // four waves take two NCHW rounds
            return patch_shape(batch, d.in_h, d.in_w, cbytes, d.out_c, nchw, g, &ps) ? g : 0;
        }
    }
    // {pixel groups, channel blocks, K parts, pixel blocks per role}.  The 7-block form: maps of 14 x 14 pixels (one image =
    // 6.1 blocks) as one round of 256 tiles without K parts -- ResNet-50 NCHW at batch 128: 256 -> 256 @14 24.2 -> 22.3 us (no
    // exchange of partial sums), 256 -> 256 @28 stride 2 39.3 -> 29.1 (its 13-block tiles were half empty).  The 4-block form
    // (7 x 7 maps) exists for SHL_MI355X_PATCH=1,4,1,4 only: 512 -> 512 @7 28.9 vs 27.5 us with four K parts, 512 -> 512 @14
    // stride 2 41.1 vs 35.1 through the re-layout passes (16 stages of nine 4-block steps are all barrier)
    // binary16 takes the 4-block form too: 512 -> 512 @14 stride 2 at batch 128 69 us against 78 (the alternative is the tile kernel
    // between two re-layout passes, 97), and at batches 8 .. 32 it fills more CUs (128 -> 128 @28 at batch 16: 15.2 us against 17.8;
    // profiles/r05_f16_nchw_small_batches.txt)
    static const int cand[][4] = {{1, 4, 1, PT_NB}, {2, 2, 1, PT_NB}, {1, 2, 2, PT_NB}, {1, 1, 4, PT_NB}, {1, 4, 1, 7}, {1, 4, 1, 4}};
    const int ocblks = (d.out_c + 31) / 32;
    int best = 0;
    double best_cost = 0;
    // binary16: a pixel is twice the bytes, and a stage's patch is bounded by what one round of staging items moves
    // (64 KB): with 128-byte stages 64 channels @56 get 3 rows per pixel group (168 of 416 pixels, 1 195 tiles); with
    // 64-byte stages 7 rows (512 tiles, two stages).  Both stage sizes are costed; ties keep the larger.
    const int kc_top = kc;
    for (int view = 0; view < 2 && !best; ++view) {
    if (view == 1) {
        if (!(f16 && nchw) || s2) break;
        nchw = false;
    }
    for (int kc_try = kc_top; kc_try >= 64; kc_try -= 64) {
    if (kc_try != kc_top && (!f16 || s2)) break;
    kc = kc_try, u = kc / 32;
    for (const auto &c : cand) {
        const int pg = c[0], ob = c[1], kp = c[2], nbt = c[3];
        if (u % kp != 0 || (s2 && kp != 1)) continue;  // (stride 2: one K sub-step per tap and stage)
        if (nbt == 4 && !f16) continue;
        if (ob > 1 && ob / 2 >= ocblks) continue;  // half of the channel blocks of a tile would be empty
        if (nbt != PT_NB && !small_tiles_ok(nchw, f16, kc, pg, ob, kp, make_geom(kc, pg, ob, kp))) continue;
        PatchShape ps;
        int g = make_geom(kc, pg, ob, kp, nbt) | (s2 ? 1 << 21 : 0) | (f16 ? 1 << 23 | 1 << 20 : 0);  // binary16: eight waves
        if (s2 && !PT_NW8(g)) return 0;  // the stride-2 form has eight waves
        if (!patch_shape(batch, d.in_h, d.in_w, cbytes, d.out_c, nchw, g, &ps)) {
            if (s2 || f16 || nbt != PT_NB) continue;
            g &= ~(1 << 20);  // four waves take two NCHW staging rounds
            if (!patch_shape(batch, d.in_h, d.in_w, cbytes, d.out_c, nchw, g, &ps)) continue;
        }
        const int64_t tiles = (int64_t)ps.nt_m * ps.nt_n;
        const double rounds = (double)((tiles + 255) / 256);
        // per tile and wave: a K loop over K / kp for the role's pixel blocks + fixed costs (prologue, epilogue; the exchange
        // of partial sums for K parts)
        const double total = rounds * ((double)nbt / PT_NB / kp + 0.12 + (kp > 1 ? 0.04 : 0.0));
        static const char *dbg = getenv("SHL_MI355X_DEBUG_GEOM");
        if (dbg) fprintf(stderr, "patch geom %d,%d,%d/%d nw%d%s: rows %d prows %d tiles %d x %d lds %d nitc %d pair %d cost %.3f\n", pg, ob, kp, nbt, PT_NW8(g) ? 8 : 4,
                         s2 ? " s2" : "", ps.rows, ps.prows, ps.nt_m, ps.nt_n, ps.lds, ps.nitc, ps.pair_dn, total);
        if (!best || total < best_cost - 1e-9) best = g, best_cost = total;
    }
    }
    }
    return best;
}

size_t patch_weight_bytes(const shl_mi355x_conv_desc &d, int geom)
{
    const int ocblks = (d.out_c + 31) / 32;
    return (size_t)ocblks * 32 * 9 * d.in_c * pt_esize(d) + 9 * 1024;  // + eight fragments of read-ahead past the last K step
}

// [channel block][K part][stage][tap][sub-step of the part][lane][16 B]: lane = (channel & 31) | (K half << 5)
void patch_pack_weights(const shl_mi355x_conv_desc &d, int geom, const int8_t *src, int8_t *dst)
{
    const int kc = PT_S2(geom) ? PT_KC(geom) / 4 : PT_KC(geom), kp = PT_KP(geom);  // (stride 2: 32-channel stages)
    // binary16: the same walk with two bytes per channel -- a K row is C * 2 bytes, a fragment piece 8 channels
    const int es = pt_esize(d);
    const int C = d.in_c * es;
    const int u = kc / 32, ui_n = u / kp, nstg = C / kc;
    const int ocblks = (d.out_c + 31) / 32;
    memset(dst, 0, patch_weight_bytes(d, geom));
    for (int ocb = 0; ocb < ocblks; ++ocb)
        for (int p = 0; p < kp; ++p)
            for (int st = 0; st < nstg; ++st)
                for (int tap = 0; tap < 9; ++tap)
                    for (int ui = 0; ui < ui_n; ++ui) {
                        int8_t *frag = dst + ((((size_t)(ocb * kp + p) * nstg + st) * 9 + tap) * ui_n + ui) * 1024;
                        const int usub = p + kp * ui;
                        for (int lane = 0; lane < 64; ++lane) {
                            const int oc = ocb * 32 + (lane & 31);
                            if (oc >= d.out_c) continue;
                            const int c0 = st * kc + usub * 32 + (lane >> 5) * 16;
                            const int ky = tap / 3, kx = tap % 3;
                            for (int b = 0; b < 16; ++b) {
                                const int cb = c0 + b, c = cb / es;  // byte of the K row, channel
                                const size_t e = d.layout == SHL_MI355X_NHWC ? (((size_t)oc * 3 + ky) * 3 + kx) * d.in_c + c
                                                                             : (((size_t)oc * d.in_c + c) * 3 + ky) * 3 + kx;
                                frag[lane * 16 + b] = src[e * es + cb % es];
                            }
                        }
                    }
}

bool patch_setup(ConvArgs &a)
{
    if (!a.w_patch || !a.pt_geom) return false;
    // the output layout follows the input's (no a.out_nchw in the kernel): conv_forward's scratch fallback hands an NHWC
    // view of an NCHW layer to kernels that store NCHW themselves -- not this one; the NHWC epilogue stores 16 channels
    // per lane
    if ((a.out_nchw != 0) != (a.in_nchw != 0)) return false;
    if (!a.in_nchw && a.Co % 16 != 0) return false;
    PatchShape ps;
    const int cbytes = a.C * (PT_F16(a.pt_geom) ? 2 : 1);
    if (PT_F16(a.pt_geom) && !PT_S2(a.pt_geom) && (a.sh != 1 || a.sw != 1)) return false;
    if (PT_S2(a.pt_geom) && (!a.in_nchw || a.sh != 2 || a.sw != 2)) return false;
    if (PT_NBT(a.pt_geom) != PT_NB && !a.in_nchw) return false;  // small tiles exist as NCHW instantiations only (patch_launch_one)
    if (!patch_shape(a.N, a.H, a.W, cbytes, a.Co, a.in_nchw != 0, a.pt_geom, &ps)) return false;
    a.pt_rows = ps.rows;
    a.pt_prows = ps.prows;
    a.pt_bufb = ps.bufb;
    a.pt_nitc = ps.nitc;
    a.pt_spr = ps.spr;
    a.pt_ntm = ps.pair_dn ? ps.nt_m / 2 : ps.nt_m;
    a.pt_pair_in = ps.pair_dn * cbytes * a.H * a.W;  // < 2^31: patch_shape
    a.pt_pair_pix = ps.pair_dn * a.H * a.W;
    const bool s2 = PT_S2(a.pt_geom);
    const int HO = s2 ? a.H / 2 : a.H, WO = s2 ? a.W / 2 : a.W;
    if (s2 && (a.Ho != HO || a.Wo != WO)) return false;
    a.pt_rW = 1.0f / (float)WO;  // the grid the patch geometry lives on = the output plane
    a.pt_rH = 1.0f / (float)HO;
    a.pt_rOW = 1.0f / (float)WO;
    a.pt_rH1 = 1.0f / (float)(HO + 1);
    a.pt_rspr = 1.0f / (float)ps.spr;
    a.pt_rntn = 1.0f / (float)ps.nt_n;
    a.pt_rHW = 1.0f / (float)(HO * WO);
    if (!a.in_nchw && !PT_F16(a.pt_geom) && (int64_t)a.M * a.Co >= 20ll << 20) a.pt_geom |= 1 << 22;  // PT_NT: large NHWC outputs past L2
    return true;
}

// vs_wave: the alternative is the barrier-free wave kernel (fewer than 128 block tiles of 128 x 128), whose time grows
// with the batch while a patch launch stays at ~one tile time: 256 -> 256 @14 at batch 32 36 us against 15
bool patch_auto(const ConvArgs &a, bool vs_wave)
{
    static const char *env = getenv("SHL_MI355X_PATCH");
    if (env && env[0] == '0') return false;
    // literal dequantise-relu-requantise epilogues stay with the older kernels (two epilogue builds here)
    const bool f16 = PT_F16(a.pt_geom);
    if (!f16 && a.act != SHL_MI355X_ACT_NONE && !a.act_clamp) return false;
    ConvArgs t = a;
    if (!patch_setup(t)) return false;
    PatchShape ps;
    patch_shape(a.N, a.H, a.W, a.C * (f16 ? 2 : 1), a.Co, a.in_nchw != 0, a.pt_geom, &ps);
    // fewer tiles than that: the latency-oriented kernels.  Measured over batches 8 .. 256 (tools/dev/batch_sweep.sh): NCHW
    // wins from 96 tiles (the alternative is a re-layout pass around another kernel), NHWC only once most CUs have a
    // tile (64 -> 64 @56 at batch 16 = 128 tiles: 10.6 us against the tile kernel's 8.4)
    const bool s2 = PT_S2(a.pt_geom);
    // (binary16: the block-tile kernels are further behind -- 256 -> 256 @14 at batch 8 26 us against 33, profiles/r04_f16_patch_kbench.txt)
    // (stride 2: from 192 -- ResNet-50's 512 -> 512 @14 is 128 tiles of 16 stages, slower than the re-layout passes around the
    // producer / consumer kernel)
    if (f16) {
        // binary16: the block-tile kernels are far behind (and an NCHW layer pays two re-layout passes around them) -- from 24 tiles, or
        // from 12 when the layer is at most four stages (a launch is ~10 us + 3 us per stage whatever the batch: 256 -> 256 @14
        // NCHW at batch 8, 16 tiles: 21.9 us against 33.1; 512 -> 512 @14 stride 2 at batch 8, 32 stages: 66 against the wave kernel's 33;
        // profiles/r05_f16_nchw_small_batches.txt)
        const int kcp = s2 ? PT_KC(a.pt_geom) / 4 : PT_KC(a.pt_geom);
        const int64_t tiles = (int64_t)ps.nt_m * ps.nt_n;
        return tiles >= 24 || (tiles >= 12 && a.C * 2 / kcp <= 4);  // (below 12 tiles: not measured, the latency kernels keep them)
    }
    if ((int64_t)ps.nt_m * ps.nt_n < ((s2 || (!a.in_nchw && !vs_wave)) ? 192 : 96)) return false;
    // NHWC with four K parts (512 channels @7 at batch 128): nine K steps per stage and the exchange of partial sums
    // leave it behind the producer / consumer kernel (25.4 vs 22.4 us); NCHW takes it anyway -- the alternative there
    // is two re-layout passes around that kernel (39 vs 47 us)
    // (binary16: 39 us against 68 through the tile kernel)
    if (!a.in_nchw && PT_KP(a.pt_geom) == 4 && !vs_wave) return false;
    return true;
}

static int g_last_tu = 0;  // translation unit of the last launch (layout | binary16 << 1): it holds the trace

int patch_launch_nhwc(const ConvArgs &a, unsigned tiles, size_t lds, hipStream_t s) { return patch_launch_layout<false>(a, tiles, lds, s); }
int patch_read_trace_nhwc(unsigned long long *host, int count) { return patch_read_trace_layout<false>(host, count); }

int patch_read_trace(unsigned long long *host, int count)
{
    switch (g_last_tu) {
        case 1: return patch_read_trace_nchw(host, count);
        case 2: return patch_read_trace_nhwc_f16(host, count);
        case 3: return patch_read_trace_nchw_f16(host, count);
        default: return patch_read_trace_nhwc(host, count);
    }
}

int launch_conv_igemm_patch(const ConvArgs &a0, hipStream_t s)
{
    ConvArgs a = a0;
    const bool f16 = PT_F16(a.pt_geom);
    if (!patch_setup(a) || (!f16 && a.act != SHL_MI355X_ACT_NONE && !a.act_clamp)) {
        set_error("conv_igemm_patch: the layer does not fit the row-patch kernel");
        return SHL_MI355X_ENOTSUP;
    }
    if (f16) a.C *= 2, a.in_zp = 0;  // the kernel's a.C is the pixel size in bytes; the padding value is 0.0
    PatchShape ps;
    patch_shape(a.N, a.H, a.W, a.C, a.Co, a.in_nchw != 0, a.pt_geom, &ps);
    const unsigned tiles = (unsigned)((ps.pair_dn ? ps.nt_m / 2 : ps.nt_m) * ps.nt_n);
    g_last_tu = (a.in_nchw != 0 ? 1 : 0) | (f16 ? 2 : 0);
    const int rc = f16 ? (a.in_nchw ? patch_launch_nchw_f16(a, tiles, ps.lds, s) : patch_launch_nhwc_f16(a, tiles, ps.lds, s))
                       : (a.in_nchw ? patch_launch_nchw(a, tiles, ps.lds, s) : patch_launch_nhwc(a, tiles, ps.lds, s));
    if (rc != SHL_MI355X_OK) return rc;
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

}  // namespace shl
