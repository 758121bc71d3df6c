// chain_xcd.hip -- a run of consecutive batch-1 layers (int8 NHWC: fused pointwise + depthwise pairs and lone
// pointwise layers) as ONE launch whose layers meet at a barrier INSIDE one XCD.
//
// Why: at batch 1 MobileNetV1 is a chain of 15 dependent launches of ~4.5 us each of which 1.6 us is the graph node
// itself and most of the rest is memory round trips nothing overlaps (profiles/r01_notes.md, r02_notes.md).  A
// device-wide barrier or flag is dearer than the kernel boundary (the eight XCDs' L2s are not coherent with each other:
// 2.6 - 11 us, tools/probes/grid_barrier.hip, flag_chain.hip).  But the workgroups of ONE XCD share one L2, and these
// layers are small enough for 32 CUs: a barrier among the workgroups of one XCD needs no L2 write-back and no L2
// invalidate -- s_waitcnt vmcnt(0) (the stores are in the L2), an atomic executed at that L2, a polling atomic.
// tools/probes/xcd_barrier.hip: 1.0 us for 16 workgroups, 1.4 us for 32, including a 1-KiB exchange.  (What the
// activation loads may assume about the L1: chain_xcd_create.)
//
//   grid = 8 G workgroups of 512 threads; workgroup i runs on XCD i % 8 (checked: HW_REG_XCC_ID is compared with the
//   first participant's and a mismatch is reported through the status word); those with i % 8 == xcd take part
//   (local id i / 8), the others leave at once.
//   per layer: items = (32-channel slice) x (rectangle of output pixels), dealt round-robin to the G workgroups;
//   the workgroup program is pwdw_body.h's (the same code as the one-launch-per-pair kernel, 8 waves).
//   between layers: arrive (after the stores) ... request the next layer's weights and per-channel constants ...
//   wait ... activation loads.  The counter only grows during a launch; the last workgroup out resets it.
// Restates the same reference functions as pwdw_fused.hip (shl_ref_conv2d_quant / shl_ref_depthwise_conv2d_quant,
// source/reference/convolution.c:370-400, 416-460, relu variants convolution_relu.c); bit-identical to the
// one-launch-per-layer path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pwdw_body.h"

namespace shl {

constexpr int CHAIN_MAX_STEPS = 32;
constexpr int CHAIN_WAVES = 8;

struct ChainStep {
    PwDwArgs f;
    int32_t items;    // slices * tiles_x * tiles_y
    int32_t slices;   // Co / 32
    int32_t tiles_x;
    int32_t pad_;
};

struct ChainArgs {
    const ChainStep *steps;  // device memory, read through the scalar cache
    uint32_t *sync;          // [0] arrivals, [1] status (1: a wait timed out, 2: participants on different XCDs), [2] first XCC id + 1
    int32_t nsteps;
    int32_t G;               // participating workgroups
    int32_t xcd;             // which i % 8 takes part
    uint32_t spin_limit;
};

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// a returning atomic is executed at the L2 whatever the L1 holds (sc0 on an atomic = "return the old value")
__device__ __forceinline__ uint32_t l2_fetch_add(uint32_t *p, uint32_t v)
{
    uint32_t r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ void l2_add(uint32_t *p, uint32_t v)
{
    asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// the wait half of the barrier: thread 0 polls the arrival counter, the other waves sit at s_barrier.  No s_waitcnt:
// the loads requested before it (weights, tables) stay in flight.
struct XcdWait {
    uint32_t *sync;
    uint32_t target;
    uint32_t spin_limit;
    bool active;
    __device__ __forceinline__ void operator()() const
    {
        if (!active) return;
        asm volatile("" ::: "memory");
        if (threadIdx.x == 0) {
            uint32_t spins = 0;
            while (l2_fetch_add(sync, 0u) < target) {
                if (++spins > spin_limit) {  // never hang the device: report and carry on
                    __hip_atomic_fetch_or(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
};

// a struct out of device memory through the scalar cache: dword loads from the constant address space
template <class T>
__device__ __forceinline__ T load_uniform(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "dword multiple");
    typedef const __attribute__((address_space(4))) uint32_t *cptr;
    cptr src = (cptr)(const void *)p;
    struct Words {
        uint32_t w[sizeof(T) / 4];
    } tmp;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) tmp.w[i] = src[i];
    return __builtin_bit_cast(T, tmp);
}

__global__ __launch_bounds__(512) void chain_xcd_kernel(ChainArgs c)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)(blockIdx.x & 7) != c.xcd) return;
    const int lid = blockIdx.x >> 3;
    const int G = c.G;
    for (int s = 0; s < c.nsteps; ++s) {
        const ChainStep st = load_uniform(c.steps + s);
        XcdWait wait{c.sync, (uint32_t)(s * G), c.spin_limit, s > 0};
        bool first = true;
        for (int item = lid; item < st.items; item += G) {
            const int rect = item / st.slices;
            const int slice = item - rect * st.slices;
            const int ty = rect / st.tiles_x;
            const int tx = rect - ty * st.tiles_x;
            if (first) {
                pwdw_body<4, 4, true>(st.f, slice, tx, ty, smem, wait);
            } else {
                __syncthreads();  // the LDS of the previous item
                pwdw_body<4, 4, true>(st.f, slice, tx, ty, smem, NoWait());
            }
            first = false;
        }
        if (first) wait();  // a workgroup without work in this layer keeps step with the others
        // arrive: every thread's stores are in the L2, then one atomic at the L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (threadIdx.x == 0) l2_add(c.sync, 1u);
    }
    // placement check, off the critical path: every participant compares its XCC id with the first one's (the word is
    // never reset: a chain keeps its XCD)
    if (threadIdx.x == 64) {
        const uint32_t me = xcc_id() + 1;
        uint32_t first = 0;
        if (!__hip_atomic_compare_exchange_strong(c.sync + 2, &first, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
            first != me)
            __hip_atomic_fetch_or(c.sync + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the last one out resets the counter for the next launch
    if (lid == 0 && threadIdx.x == 0) {
        uint32_t spins = 0;
        while (l2_fetch_add(c.sync, 0u) < (uint32_t)(c.nsteps * G)) {
            if (++spins > c.spin_limit) {
                __hip_atomic_fetch_or(c.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(c.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- host side ---------------------------------------------------------------------------------
struct XcdChain {
    ChainStep *steps_dev = nullptr;
    uint32_t *sync_dev = nullptr;
    ChainStep steps[CHAIN_MAX_STEPS];
    int nsteps = 0;
    int G = 32;
    int xcd = 0;
    size_t lds = 0;
    int device = 0;
};

static int chain_workgroups()
{
    const char *env = getenv("SHL_MI355X_CHAIN_WGS");  // 8 .. 32 (A/B); one workgroup per CU of the XCD
    const int g = env ? atoi(env) : 32;
    return g < 1 ? 1 : (g > 32 ? 32 : g);
}

// a lone pointwise layer as a step: rectangles of full-width rows, no halo
static bool choose_rows(const ConvArgs &q, PwDwArgs &f, int cus)
{
    const int nsub = q.C >> 5;
    const int ks = nsub > 16 ? 8 : (nsub >= 8 ? 4 : (nsub >= 4 ? 2 : 1));
    const int nsw = (nsub + ks - 1) / ks;
    if (nsw > 4) return false;
    const int mt_max = (CHAIN_WAVES / ks) * 4;
    const int64_t slices = q.Co >> 5;
    double best = 1e30;
    int best_h = 0;
    for (int bh = 1; bh <= q.H; ++bh) {
        const int npx = bh * q.W;
        const int mt = (npx + 31) / 32;
        if (mt > mt_max || npx >= 4096) break;
        if ((size_t)mt * ks * 4096 > (size_t)156 * 1024) break;
        const int64_t blocks = slices * ((q.H + bh - 1) / bh);
        const double rounds = (double)((blocks + cus - 1) / cus);
        const double score = rounds * ((mt + 1) * nsub + 4.0) - (blocks <= cus ? blocks / 1024.0 : 0.0);
        if (score < best) best = score, best_h = bh;
    }
    if (!best_h) return false;
    f.bh = best_h;
    f.bw = q.W;
    f.tiles_y = (q.H + f.bh - 1) / f.bh;
    f.tiles_x = 1;
    f.rw = q.W;
    f.npx = f.bh * q.W;
    f.mt = (f.npx + 31) / 32;
    f.nwaves = CHAIN_WAVES;
    f.ks = ks;
    f.nsw = nsw;
    f.nsub = nsub;
    f.rw_magic = ((1u << 20) + f.rw - 1) / f.rw;
    f.bw_magic = f.rw_magic;
    f.pw_only = 1;
    return true;
}

static bool pointwise_ok(const ConvArgs &q)
{
    if (q.Kh != 1 || q.Kw != 1 || q.sh != 1 || q.sw != 1 || q.pt != 0 || q.pl != 0) return false;
    if (q.H != q.Ho || q.W != q.Wo || (q.C & 31) != 0 || (q.Co & 31) != 0 || q.kstride < q.C) return false;
    if (q.C > 1024 || q.N != 1) return false;
    if ((int64_t)q.H * q.W * q.C >= ((int64_t)1 << 31)) return false;
    return true;
}

static bool make_step(const ConvArgs &q, const ConvArgs *d, int G, ChainStep &st)
{
    memset(&st, 0, sizeof(st));
    PwDwArgs &f = st.f;
    f.pw = q;
    if (d) {
        f.dw = *d;
        if (q.N != 1 || !pwdw_shapes_pair(q, *d) || !pwdw_choose_rect(q, *d, f, CHAIN_WAVES, G)) return false;
        if (f.nsw > 4) return false;
    } else {
        if (!pointwise_ok(q) || !choose_rows(q, f, G)) return false;
    }
    st.slices = q.Co >> 5;
    st.tiles_x = f.tiles_x;
    st.items = st.slices * f.tiles_x * f.tiles_y;
    return true;
}

bool chain_xcd_unit_ok(const ConvArgs &q, const ConvArgs *d, int pw_is_igemm, int dw_dot4_packed)
{
    if (!pw_is_igemm || (d && !dw_dot4_packed)) return false;
    ChainStep st;
    return make_step(q, d, chain_workgroups(), st);
}

int chain_xcd_create(const ChainUnit *units, int n, XcdChain **out)
{
    static int next_xcd = 0;
    if (n < 1 || n > CHAIN_MAX_STEPS) {
        set_error("chain_xcd: 1 .. 32 units");
        return SHL_MI355X_EINVAL;
    }
    XcdChain *c = new XcdChain;
    c->G = chain_workgroups();
    c->nsteps = n;
    for (int i = 0; i < n; ++i) {
        if (!make_step(units[i].pw, units[i].has_dw ? &units[i].dw : nullptr, c->G, c->steps[i])) {
            delete c;
            set_error("chain_xcd: a unit does not qualify");
            return SHL_MI355X_ENOTSUP;
        }
        const PwDwArgs &f = c->steps[i].f;
        const size_t lds = (size_t)f.mt * f.ks * 4096 + (f.pw_only ? 0 : (size_t)f.mt * 1024);
        c->lds = lds > c->lds ? lds : c->lds;
    }
    // The activations are read with ordinary loads (through the L1): a compute unit's L1 is invalidated when the launch
    // starts, takes in a line only when it loads it, and every tensor is loaded only after the barrier that follows its
    // one and only writer -- so no L1 can hold a stale line PROVIDED no address is written after it was read inside
    // the launch: all tensors of the chain disjoint, on their own 128-byte lines.  (Reading past the L1 instead -- sc1
    // loads, tools/probes/xcd_barrier.hip -- works for any buffer assignment but quadruples the L2 requests of the
    // 16-byte fragment loads: 7.2 us per layer against 4.6 us for the stand-alone launches.)
    {
        struct Span {
            uintptr_t lo, hi;
        } spans[CHAIN_MAX_STEPS + 1];
        int ns = 0;
        for (int i = 0; i < n; ++i) {
            const ConvArgs &q = units[i].pw;
            const ConvArgs &o = units[i].has_dw ? units[i].dw : units[i].pw;
            const uintptr_t in = (uintptr_t)q.in, out = (uintptr_t)o.out;
            if (i == 0) spans[ns++] = {in, in + (uintptr_t)q.H * q.W * q.C};
            else if (in != spans[ns - 1].lo) {
                delete c;
                set_error("chain_xcd: unit %d does not read what unit %d writes", i, i - 1);
                return SHL_MI355X_EINVAL;
            }
            spans[ns++] = {out, out + (uintptr_t)o.Ho * o.Wo * o.Co};
        }
        for (int a = 0; a < ns; ++a) {
            bool bad = (spans[a].lo & 127) != 0;
            for (int b = 0; b < a; ++b) bad = bad || (spans[a].lo < ((spans[b].hi + 127) & ~(uintptr_t)127) && spans[b].lo < ((spans[a].hi + 127) & ~(uintptr_t)127));
            if (bad) {
                delete c;
                set_error("chain_xcd: the tensors of a chain must be disjoint and 128-byte aligned (tensor %d)", a);
                return SHL_MI355X_ENOTSUP;
            }
        }
    }
    const char *env = getenv("SHL_MI355X_CHAIN_XCD");
    c->xcd = env ? (atoi(env) & 7) : (next_xcd++ & 7);  // concurrent chains (sessions on their own streams) spread over the XCDs
    SHL_HIP(hipGetDevice(&c->device));
    SHL_HIP(hipMalloc(&c->steps_dev, sizeof(ChainStep) * n));
    SHL_HIP(hipMalloc(&c->sync_dev, 256));
    SHL_HIP(hipMemcpy(c->steps_dev, c->steps, sizeof(ChainStep) * n, hipMemcpyHostToDevice));
    SHL_HIP(hipMemset(c->sync_dev, 0, 256));
    *out = c;
    return SHL_MI355X_OK;
}

int chain_xcd_launch(XcdChain *c, hipStream_t s)
{
    ChainArgs a;
    a.steps = c->steps_dev;
    a.sync = c->sync_dev;
    a.nsteps = c->nsteps;
    a.G = c->G;
    a.xcd = c->xcd;
    a.spin_limit = 1u << 20;  // ~ a second
    static LdsOptIn opted_in;
    if (c->lds > 64 * 1024) lds_opt_in(opted_in, reinterpret_cast<const void *>(chain_xcd_kernel));
    hipLaunchKernelGGL(chain_xcd_kernel, dim3(8 * c->G), dim3(64 * CHAIN_WAVES), c->lds, s, a);
    SHL_HIP(hipGetLastError());
    return SHL_MI355X_OK;
}

int chain_xcd_status(XcdChain *c, uint32_t *status)
{
    uint32_t w[4] = {0, 0, 0, 0};
    SHL_HIP(hipMemcpy(w, c->sync_dev, sizeof(w), hipMemcpyDeviceToHost));
    *status = w[1];
    return SHL_MI355X_OK;
}

void chain_xcd_describe(const XcdChain *c, int step, int32_t *out8)
{
    const ChainStep &st = c->steps[step];
    out8[0] = st.items, out8[1] = st.slices, out8[2] = st.f.bh, out8[3] = st.f.bw, out8[4] = st.f.mt, out8[5] = st.f.ks;
    out8[6] = st.f.pw_only, out8[7] = c->G;
}

void chain_xcd_destroy(XcdChain *c)
{
    if (!c) return;
    if (c->steps_dev) hipFree(c->steps_dev);
    if (c->sync_dev) hipFree(c->sync_dev);
    delete c;
}

}  // namespace shl
