// render.hip -- per-tile compositing kernels for gfx950 (CDNA4, wave64):
//   F6  front-to-back alpha compositing -> color, depth, alpha (+ final_T, n_contrib)
//         render_fwd_wave_kernel   one wave per (tile, 8x8 quadrant): batches of views (the chip is full)
//         render_fwd_seg_kernel    eight waves per (tile, quadrant), segment-parallel: one or two views (the chip is not)
//   B1  gradients of the compositing -> one partial record per (tile instance, quadrant)   (render_bwd_bucket_kernel)
// Replaces renderCUDA forward/backward of the third-party rasterizer behind
// /root/reference/core/gaussians/gs.py:98-106 and train_vae.py:166, for all views of a batch in ONE launch
// (the reference issues B*V separate launch chains, gs.py:62,75).
//
// A wave owns an 8x8 quadrant of a 16x16 tile, so that its 64 pixels are spatially compact: early termination and
// sub-tile culling are decided per wave.  The tile list is culled for the quadrant with a ballot over an exact
// per-Gaussian bounding test (the half extents of the alpha >= 1/255 ellipse, written by preprocess): culled
// Gaussians would have hit the published `alpha < 1/255 -> continue` rule at every pixel of the quadrant, so results
// (including n_contrib) are unchanged by the cull.
//
// THE ALPHA TEST.  The published rule `alpha = min(0.99, opacity * exp(power)); if (alpha < 1/255) continue` is a
// threshold on the exponent: alpha is monotone in `power`, so for every Gaussian there is ONE float p* with
//     alpha >= 1/255   <=>   power >= p*.
// preprocess.hip computes p* per Gaussian with a correctly rounded exp2 (fp64, the same sequence of IEEE operations as
// the CPU oracle) and stores it in the record (its twelfth float); the kernels below test `p* <= power <= 0` -- the same instruction count
// as comparing alpha, but the decision no longer depends on the last bit of v_exp_f32: every forward kernel, the
// backward and the CPU oracle take the same decision at every (pixel, Gaussian) pair, bit for bit.
//
// Roofline: algorithmic HBM bytes are 44 B per tile instance + 24 B (fwd) / 28 B (bwd) per pixel
// (SURVEY.md 8d); with 3-4 px splats the inner loop is VALU-bound, not HBM-bound -- see DESIGN.md.
#include <cstdlib>
#include <string.h>
#include "common.h"

namespace {

#ifndef SGR_FWD_G
#define SGR_FWD_G 2            // Gaussians per iteration of the wave forward: 2 -> 70 VGPRs (7 waves/SIMD); 4 is 12% slower at 64 views
#endif
constexpr float kLog2e = -1.4426950408889634f;          // conic.xy is pre-multiplied by -log2(e)
constexpr float kHalfLog2e = -0.7213475204444817f;      // conic.xx / conic.yy by -0.5*log2(e)
constexpr float kNever = __builtin_inff();              // p* of an entry that must never pass the alpha test (null padding)

// bounding test of the alpha >= 1/255 ellipse against the quadrant whose first pixel is (qx0, qy0)
// (c.z = the two half extents as bf16 halves of one word, rounded up by preprocess: hx in the low half, hy in the high half; c.w = p*)
__device__ __forceinline__ bool cull_quadrant(const float4 &a, const float4 &c, float qx0, float qy0) {
    const uint32_t h2 = __float_as_uint(c.z);
    const float hx = __uint_as_float(h2 << 16), hy = __uint_as_float(h2 & 0xFFFF0000u);
    if (hx < 0.f) return false;                    // opacity <= 1/255: can never pass the alpha floor
    return (a.x + hx >= qx0) && (a.x - hx <= qx0 + 7.f) && (a.y + hy >= qy0) && (a.y - hy <= qy0 + 7.f);
}

// auxiliary forward outputs that feed the bucket-parallel backward (all optional; see sgr_render_forward)
struct FwdAux {
    uint2 *compact;       // [4][R]  per (tile, quadrant) culled list in order: (record id, 0-based index in the tile list)
    // Checkpoints [4*NS][4][64]: per bucket of 64 survivors, per 16-survivor row, per pixel: (T, C0, C1, C2) before the row's first
    // survivor.  T is absolute.  With rps = rows per forward segment (1..4, in the descriptor): rows with r % rps == 0 start a
    // segment and hold ABSOLUTE composited sums, the others hold the sums accumulated since their segment's start (row r - r % rps).
    // The row at ordinal 0 of a list is never stored (T = 1, sums = 0).
    // (Round 2 also had a "compact" layout -- one checkpoint per bucket, the inner rows rebuilt by the backward: 4x smaller, backward
    // +55 % -- selected when the rows would not fit 8 GiB; no BASELINE config came near that, removed in round 5.)
    float4 *ckpt_tc;      // NULL: "depth/alpha only" pass (see ckpt_da): nothing but ckpt_da is written, no outputs either
    float2 *ckpt_da;      // same for (D, A).  NULL: not stored -- only a backward with dL/ddepth or dL/dalpha reads them (never on the reference's
                          // call paths), so by default they are produced on demand by a second compositing pass with ckpt_tc = NULL
    uint2 *desc;          // [4*NS]  (global tile id | (rps - 1) << 30, (start << 7) | count): `count` (<= 64) survivors starting at ordinal `start`
                          //          (a multiple of 64) of the (tile, quadrant) list; count == 0 -> slot unused
    uint32_t R, NS;
    float *clamped;       // optional [n_views,3,H,W]: clamp(colour, 0, 1) next to the unclamped colours (SgrProblem.color_clamped: gs.py:107 folded in)
};

// The single-view FUSED step (sgr_rasterize_forward_l1 with SgrL1Epilogue.fuse_backward; AUX >= 3 of the segment-parallel kernel): the clamp +
// masked L1 of gs.py:107 / whole_loss.py:126-131 is pixel-local, so the workgroup that composited a (tile, quadrant) knows the loss share and
// dL/dcolor of its 64 pixels the moment its forward is over: no loss launch (AUX == 3; the bucket backward is queued right behind by the same
// host call and sums the loss shares on the side).  (Running the backward of the workgroup's OWN buckets on the spot as well -- no backward
// launch either -- was built and measured: slower at C1 and C2, DESIGN.md dead ends (aj).)
struct FusedL1 {
    const float *target;      // [n_views,3,H,W]
    const float *mask;        // [n_views,1,H,W] or NULL
    float weight;
    float *gimg;              // [n_views,3,H,W]  dL/dcolor (written like clamped_l1_kernel does: a backward that is handed another upstream gradient starts from it)
    float *loss_part;         // [n_views*tiles*4]  weight * sum |clamp(colour) - target| * mask per (tile, quadrant), summed in a fixed order (l1_reduce_block)
};

// Sums of products are written with their fused multiply-adds spelled out.  `a*b + c*d` may be contracted with either product inside
// the fma, the compiler picks by operand arrival, and the pick differed between kernels (round 3: the wave forward fused the kyy term
// of the exponent, the segment-parallel forward and the backward the kxy term; the backward's colour dot changed its order when its LDS
// reads moved).  Written out, every forward kernel, the backward AND the CPU oracle (its ref_power2) agree TO THE BIT
// on a Gaussian's exponent at a pixel.
__device__ __forceinline__ float sgr_power2(float kxx, float kyy, float kxy, float dx, float dy) {   // (exp2 domain: conic pre-scaled)
    return fmaf(dx, kxx * dx, fmaf(kxy * dx, dy, (kyy * dy) * dy));
}
__device__ __forceinline__ float sgr_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}
// -------------------------------------------------------------------------------------------------
// F6, one WAVE per (tile, quadrant) -- the default for launches that fill the chip (batches of views).  Every quadrant is its own
// 64-thread workgroup: it walks the tile list in batches of 64 entries (lane j stages entry j: record prefetched one batch ahead, ids
// two ahead), culls them for its own 8x8 pixels with one ballot, and composites the survivors -- no workgroup
// barrier anywhere, a quadrant whose pixels have all stopped ends at once and frees its SIMD slot.  The four waves of a tile re-read the
// tile's records (L2 hits: workgroups b, b+8, b+16, b+24 are the four quadrants of one tile and run on the same XCD).
// -------------------------------------------------------------------------------------------------
constexpr int kWaveBatch = 64;
#ifndef SGR_WAVE_AUX_WAVES
#define SGR_WAVE_AUX_WAVES 7   // waves per SIMD asked for the checkpointing instantiation: 7 = 68 VGPRs without spills; 8 = 64 VGPRs + 16 B of scratch
#endif
template <int AUX>
__global__ __launch_bounds__(64, AUX ? SGR_WAVE_AUX_WAVES : 8) void render_fwd_wave_kernel(int W, int H, int Tx, uint32_t tiles_per_view, uint32_t tiles_total,
                                                             const uint2 *__restrict__ ranges,
                                                             const uint32_t *__restrict__ point_list,
                                                             const float4 *__restrict__ rec, const float *__restrict__ bg,
                                                             float *__restrict__ out_color, float *__restrict__ out_depth,
                                                             float *__restrict__ out_alpha, float *__restrict__ final_T,
                                                             uint32_t *__restrict__ n_contrib, FwdAux aux) {
    __shared__ float4 sA[kWaveBatch + 1], sB[kWaveBatch + 1], sC[kWaveBatch + 1];   // entry 64 = null Gaussian (opacity 0)
    __shared__ uint16_t sJ[kWaveBatch];                                             // batch-local entry index of survivor g
    const uint32_t logical = (blockIdx.x >> 5) * 8u + (blockIdx.x & 7u), q = (blockIdx.x >> 3) & 3u;
    if (logical >= tiles_total) return;
    const uint32_t bid = sgr_xcd_remap(logical, tiles_total);
    const uint32_t view = bid / tiles_per_view, tile = bid - view * tiles_per_view;
    const uint32_t tx = tile % Tx, ty = tile / Tx;
    const uint2 range = ranges[bid];
    const int lane = threadIdx.x;
    const int px = (int)tx * 16 + (int)(q & 1u) * 8 + (lane & 7);
    const int py = (int)ty * 16 + (int)(q >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float qx0 = (float)(tx * 16 + (q & 1u) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0, lastk = 0;
    uint32_t kbase = 0;                                  // survivors of this quadrant in earlier batches
    const int n = (int)(range.y - range.x);
    const size_t slot0 = (size_t)q * aux.NS + (range.x >> 6) + (size_t)bid;            // first bucket slot of (tile, quadrant)
    // software pipeline: (ra, rb, rc, rid) = this lane's entry of the CURRENT batch, id_nx = its entry id of the next batch
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = make_float4(0.f, 0.f, -1.f, -1.f);
    uint32_t rid = 0, id_nx = 0;
    if (lane < n) { rid = point_list[range.x + lane]; ra = rec[(size_t)rid * 4 + 0]; rb = rec[(size_t)rid * 4 + 1]; rc = rec[(size_t)rid * 4 + 2]; }
    if (lane + kWaveBatch < n) id_nx = point_list[range.x + lane + kWaveBatch];
    for (int base = 0; base < n; base += kWaveBatch) {
        if (!__builtin_amdgcn_ballot_w64(!done)) break;
        // ---- stage the current batch (conic pre-scaled into the exp2 domain: exp(power) = exp2(kxx dx^2 + kyy dy^2 + kxy dx dy)) and
        // cull it for this quadrant
        // -- only the survivors go to LDS, compacted: survivor g of the batch sits at index g (no index list to read back)
        const bool bit = (base + lane < n) && cull_quadrant(ra, rc, qx0, qy0);
        const uint64_t bal = __ballot(bit);
        const uint32_t cnt = (uint32_t)__popcll(bal);
        const uint32_t pos = (uint32_t)__popcll(bal & lt_mask);
        if (bit) {
            sA[pos] = make_float4(ra.x, ra.y, kHalfLog2e * ra.z, kLog2e * ra.w);
            sB[pos] = make_float4(kHalfLog2e * rb.x, rb.y, rb.z, rb.w);
            sC[pos] = rc;                                               // (g, b, -, p*)
            sJ[pos] = (uint16_t)lane;
            if (AUX && aux.ckpt_tc) aux.compact[(size_t)q * aux.R + range.x + kbase + pos] = make_uint2(rid, (uint32_t)(base + lane));
        }
        if (lane == 0) { sA[cnt] = make_float4(0.f, 0.f, 0.f, 0.f); sB[cnt] = sA[cnt]; sC[cnt] = make_float4(0.f, 0.f, 0.f, kNever); }      // null Gaussian behind an odd count: never valid
        // ---- next batch: records now, ids of the batch after it
        {
            const int nx = base + kWaveBatch + lane;
            rid = id_nx;
            rc = make_float4(0.f, 0.f, -1.f, -1.f);
            if (nx < n) { ra = rec[(size_t)rid * 4 + 0]; rb = rec[(size_t)rid * 4 + 1]; rc = rec[(size_t)rid * 4 + 2]; }
            if (nx + kWaveBatch < n) id_nx = point_list[range.x + nx + kWaveBatch];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t lastg = 0xFFFFFFFFu;                                   // survivor of this batch that contributed last to my pixel
        uint64_t active = __builtin_amdgcn_ballot_w64(!done);
        // AUX: bit g = a checkpoint is due in front of survivor g of this batch -- its ordinal kbase + g is a multiple of 16 and not 0.  One scalar mask per batch instead of three scalar compares per survivor (the walk is a serial chain per
        // wave: every instruction in it is latency; the checkpoint bookkeeping was 8 of its 22 scalar instructions per survivor).
        uint64_t ck = 0;
        if constexpr (AUX != 0) ck = 0x0001000100010001ull << ((0u - kbase) & 15u);
        if constexpr (AUX != 0) {
            if (kbase == 0u) ck &= ~1ull;
            if (cnt < 64u) ck &= (1ull << cnt) - 1ull;
        }
        // "every pixel is done" is looked at every fourth group only: turning the compiler's lane mask of `done` into a loop condition costs two
        // vector and five scalar instructions; the (at most three) groups composited after the last pixel stopped change nothing (w = 0)
        for (uint32_t g = 0; g < cnt && active;) {
          const uint32_t g_end = min(cnt, g + 4u * SGR_FWD_G);
          for (; g < g_end; g += SGR_FWD_G) {
            float4 a[SGR_FWD_G], b[SGR_FWD_G], c[SGR_FWD_G];
            float al[SGR_FWD_G];
            bool valid[SGR_FWD_G];
#pragma unroll
            for (int u = 0; u < SGR_FWD_G; u++) { a[u] = sA[g + u]; b[u] = sB[g + u]; c[u] = sC[g + u]; }
#pragma unroll
            for (int u = 0; u < SGR_FWD_G; u++) {
                const float dx = a[u].x - pxf, dy = a[u].y - pyf;
                const float power = sgr_power2(a[u].z, b[u].x, a[u].w, dx, dy);
                const float alpha = fminf(0.99f, b[u].y * __builtin_amdgcn_exp2f(power));
                valid[u] = (power <= 0.f) & (power >= c[u].w);            // p* <= power <= 0  <=>  the published alpha test (see the file header)
                al[u] = valid[u] ? alpha : 0.f;
            }
            // sequential part, branch-free: the only loop-carried chain is T -> test_T -> (stop) -> T
            const uint32_t due = AUX ? (uint32_t)(ck >> g) & ((1u << SGR_FWD_G) - 1u) : 0u;
#pragma unroll
            for (int u = 0; u < SGR_FWD_G; u++) {
                const uint32_t ord = kbase + g + u;                       // ordinal of this survivor in the quadrant list
                if constexpr (AUX != 0) {
                    if (__builtin_expect((due >> u) & 1u, 0u)) {
                        // every row holds the ABSOLUTE state (descriptor: one row per segment); the segment-parallel kernel can only give
                        // its inner rows relative to their segment's start
                        const size_t s = ((slot0 + (ord >> 6)) * 4 + ((ord >> 4) & 3u)) * 64 + lane;
                        if (aux.ckpt_tc) aux.ckpt_tc[s] = make_float4(T, C0, C1, C2);
                        if (aux.ckpt_da) aux.ckpt_da[s] = make_float2(D, A);
                    }
                }
                // al = 0 for a Gaussian that is not valid here: its test_T is T itself (>= 1e-4 while the pixel is not done), its weight 0 and
                // its new T the old one -- only `done` has to gate the updates, `valid` only the index of the last contributor
                const float test_T = T * (1.f - al[u]);
                done = done | (test_T < 0.0001f);                         // the crossing Gaussian is NOT composited
                const float w = done ? 0.f : al[u] * T;
                C0 = fmaf(b[u].w, w, C0); C1 = fmaf(c[u].x, w, C1); C2 = fmaf(c[u].y, w, C2);
                D = fmaf(b[u].z, w, D);
                A += w;
                T = done ? T : test_T;
                lastg = (valid[u] & !done) ? g + u : lastg;
            }
          }
          active = __builtin_amdgcn_ballot_w64(!done);
        }
        if (lastg != 0xFFFFFFFFu) {
            last = (uint32_t)base + sJ[lastg] + 1u;
            if (AUX) lastk = kbase + lastg + 1u;                          // ordinal + 1 of the last survivor that blended into my pixel
        }
        kbase += cnt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");              // the next batch overwrites the staging arrays
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (inside && !(AUX && !aux.ckpt_tc)) {
        const size_t hw = (size_t)H * W;
        const size_t pix = (size_t)py * W + px;
        const size_t vb = (size_t)view * hw;
        final_T[vb + pix] = T;
        n_contrib[vb + pix] = last;
        const float o0 = C0 + T * bg[0], o1 = C1 + T * bg[1], o2 = C2 + T * bg[2];
        out_color[(vb * 3) + pix] = o0;
        out_color[(vb * 3) + hw + pix] = o1;
        out_color[(vb * 3) + 2 * hw + pix] = o2;
        if (aux.clamped) {
            aux.clamped[(vb * 3) + pix] = fminf(fmaxf(o0, 0.f), 1.f);
            aux.clamped[(vb * 3) + hw + pix] = fminf(fmaxf(o1, 0.f), 1.f);
            aux.clamped[(vb * 3) + 2 * hw + pix] = fminf(fmaxf(o2, 0.f), 1.f);
        }
        out_depth[vb + pix] = D;
        out_alpha[vb + pix] = A;
    }
    if (AUX) {
        uint32_t kmax = lastk;                                 // survivors up to the last one that blended anywhere in the quadrant
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        const uint32_t nb = (kmax + 63u) >> 6;
        for (uint32_t bk = lane; bk < nb && aux.ckpt_tc; bk += 64)
            aux.desc[slot0 + bk] = make_uint2(bid, (bk << 13) | min(64u, kmax - (bk << 6)));                // rps = 1: every row absolute
    }
}

// -------------------------------------------------------------------------------------------------
// F6, segment-parallel variant for launches that cannot fill the chip (one 512^2 humanoid view has ~200 occupied
// tiles for 256 CUs, and the serial walk of the longest tile list -- ~2900 entries at C2 -- is the whole critical path).
// One workgroup = one (tile, quadrant), 8 waves.  The tile list is streamed in 512-entry sub-chunks (software
// pipelined: ids two sub-chunks ahead, records one ahead, both held in registers), culled for this quadrant and
// appended to an LDS ring.  Whenever the ring holds >= 512 survivors (or the list is exhausted) one ROUND composites
// up to 512 of them as 8 contiguous segments of 64 (one per wave):
//   phase 1  every wave computes its segment's transmittance product per pixel (alpha only),
//   prefix   T_in(segment) = T_carry * prod(earlier segments)       (same association in every wave),
//   phase 2  every wave composites its segment with the published sequential rule starting from T_in.
// Contributions are absolute (already multiplied by T), so the 8 partial sums simply add up at the end.
// A pixel that stopped in an earlier segment has T_in < 1e-4 (T is monotone), so later segments skip it.
// ~1.5x the arithmetic of the serial kernel, 8x shorter dependency chain.  Each segment doubles as one bucket of
// the bucket-parallel backward with its checkpoint (T_in, composited-so-far); cutting segments at exactly 64
// survivors (only the last one of a list is shorter) keeps all four 16-lane rows of the backward's waves busy.
// -------------------------------------------------------------------------------------------------
// (waves per workgroup, measured in round 5 on one box: 4 -> C2 forward 48.8 -> 61.8 us, C1 26.5 -> 23.5; 16 -> one workgroup per CU and
// spills, C2 87 us.  The heaviest tile's chain of rounds is the critical path at C2, the number of resident workgroups at C1.)
#ifndef SGR_SEG_WAVES
#define SGR_SEG_WAVES 8
#endif
constexpr int kSegWaves = SGR_SEG_WAVES, kSegThreads = 64 * kSegWaves, kSegRing = 2 * kSegThreads, kSegPer = 64;
typedef float v2f __attribute__((ext_vector_type(2)));      // <2 x float>: the backend selects v_pk_{add,mul,fma}_f32 for it


#ifdef SGR_SEG_TRACE
// Dev instrumentation (tools/seg_trace.py, built by tools/build_ab.sh with -DSGR_SEG_TRACE): wave 0 of the workgroups of the first
// kSegTraceSlots work-order slots records (s_memtime << 8 | event id) at every phase boundary.  Not compiled into the product library.
constexpr int kSegTraceSlots = 8, kSegTraceEvents = 512, kSegTraceSched = kSegTraceSlots * 4 * kSegTraceEvents;   // schedule records start here: 8 words per (slot, q)
__device__ unsigned long long *g_seg_trace = nullptr;
extern "C" int sgr_debug_seg_trace(void *buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_seg_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
#define SGR_TR(ID) do { if (tr_buf && tr_n < kSegTraceEvents) { tr_buf[tr_n++] = ((unsigned long long)__builtin_readcyclecounter() << 8) | (unsigned)(ID); } } while (0)
#else
#define SGR_TR(ID) do { } while (0)
#endif

template <int AUX>
__global__ __launch_bounds__(kSegThreads, 4) void render_fwd_seg_kernel(int W, int H, int Tx, uint32_t tiles_per_view,
                                                                     const uint2 *__restrict__ ranges,
                                                                     const uint32_t *__restrict__ point_list,
                                                                     const float4 *__restrict__ rec, const float *__restrict__ bg,
                                                                     float *__restrict__ out_color, float *__restrict__ out_depth,
                                                                     float *__restrict__ out_alpha, float *__restrict__ final_T,
                                                                     uint32_t *__restrict__ n_contrib, FwdAux aux,
                                                                     const uint32_t *__restrict__ order, uint32_t n_slots, FusedL1 fz, int bg_done, int order_kind) {
    // survivor ring, stored as PAIRS of consecutive survivors with the two survivors' values of each field adjacent, so that the
    // per-pixel arithmetic of both runs as packed fp32 (v_pk_*: two survivors per instruction) straight out of ds_read_b128:
    __shared__ float4 pA[kSegRing / 2], pB[kSegRing / 2], pC[kSegRing / 2];   // (x0,x1,y0,y1) (kxx0,kxx1,kxy0,kxy1) (kyy0,kyy1,op0,op1)
    __shared__ float4 pD[kSegRing / 2], pE[kSegRing / 2];                     // (r0,g0,r1,g1) (b0,depth0,b1,depth1): per-survivor channel pairs
    __shared__ float4 pQ[kSegRing / 2];                                       // (p*0, p*1, bits of (list index + 1) of both): phase 1 reads the first half
    __shared__ float sT[kSegWaves][64];
    __shared__ float sAcc[kSegWaves][5][64];
    __shared__ float sTstop[64];
    __shared__ uint32_t sLast[kSegWaves][64];
    __shared__ uint32_t sWaveCnt[kSegWaves];
    __shared__ uint32_t sContrib[kSegWaves];
    __shared__ float sTgt[AUX >= 3 ? 4 : 1][64];                             // FUSED loss: the pixels' target colour and mask, fetched straight into LDS at the start
    // work order: longest lists first (fwd_prepare_kernel); the tail of the grid are the empty tiles, which only write the background
    // XCD placement: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the four quadrant workgroups
    // of a tile -- which gather the same records -- are given ids b, b+8, b+16, b+24: same XCD, dispatched back to back
    const uint32_t slot = (blockIdx.x >> 5) * 8u + (blockIdx.x & 7u);
    const uint32_t q = (blockIdx.x >> 3) & 3u;
    if (slot >= n_slots) return;                                 // (grid padded to a multiple of 32)
    uint32_t bid = slot;
    uint2 range;
    if (order_kind == 1) {
        // class-major order (common.h SGR_ORDER_HDR_WORDS): slot k is the k-th occupied tile counting through the classes, longest lists first;
        // slots beyond the occupied tiles have nothing to do (the empty tiles' background was written by the per-tile sort launch)
        uint32_t k = slot, cls = 0u;
        bool found = false;
#pragma unroll
        for (uint32_t c = 0; c < 32u; c++) {
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)order[c]);
            if (!found) { if (k < cnt) { found = true; cls = c; } else k -= cnt; }
        }
        if (!found) return;
        const uint4 od = reinterpret_cast<const uint4 *>(order + SGR_ORDER_HDR_WORDS)[(size_t)cls * n_slots + k];
        bid = od.x; range = make_uint2(od.y, od.z);
    } else if (order) { const uint4 od = reinterpret_cast<const uint4 *>(order)[slot]; bid = od.x; range = make_uint2(od.y, od.z); }   // (tile, its range): one load
    else range = ranges[bid];
    const uint32_t view = bid / tiles_per_view, tile = bid - view * tiles_per_view;
    const uint32_t tx = tile % Tx, ty = tile / Tx;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
#ifdef SGR_SEG_TRACE
    unsigned long long *tr_buf = (g_seg_trace && slot < (uint32_t)kSegTraceSlots && t == 0) ? g_seg_trace + ((size_t)slot * 4 + q) * kSegTraceEvents : nullptr;
    int tr_n = 2;
    if (tr_buf) { tr_buf[1] = ((unsigned long long)(uint32_t)(range.y - range.x) << 32) | bid; }
    SGR_TR(1);
    unsigned long long *sch = (g_seg_trace && t == 0) ? g_seg_trace + kSegTraceSched + ((size_t)slot * 4 + q) * 8 : nullptr;
    if (sch) {
        sch[0] = wall_clock64();
        sch[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        sch[3] = ((unsigned long long)(uint32_t)(range.y - range.x) << 32) | bid;
        sch[1] = sch[0];
        sch[4] = 0ull; sch[5] = 0ull;
    }
#endif
    const int px = (int)tx * 16 + (int)(q & 1u) * 8 + (lane & 7);
    const int py = (int)ty * 16 + (int)(q >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float qx0 = (float)(tx * 16 + (q & 1u) * 8), qy0 = (float)(ty * 16 + (q >> 1) * 8);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int n = (int)(range.y - range.x);
    if (n == 0) {
        // An empty tile (812 of the 1024 tiles of a 512^2 humanoid view) only receives the background.  Its quadrant 0 workgroup writes all
        // 256 pixels, the other three leave at once: these workgroups are dispatched behind the working ones and each holds a full
        // workgroup's registers and LDS while it lives -- four of them per empty tile, each walking through the whole kernel with 64 of its
        // 512 threads writing, were 5 of the kernel's 53 us at C2.  (Workgroup-uniform exit in front of the first barrier.)
        // (bg_done: extra workgroups of the preprocess launch already did all of it -- SgrBgJob, common.h)
        if (bg_done || q != 0u || (AUX && !aux.ckpt_tc) || t >= 256) return;
        const int bx = (int)tx * 16 + (t & 15), by = (int)ty * 16 + (t >> 4);
        if (bx < W && by < H) {
            const size_t hw = (size_t)H * W, pix = (size_t)by * W + bx, vb = (size_t)view * hw;
            const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
            final_T[vb + pix] = 1.f;
            n_contrib[vb + pix] = 0u;
            out_color[(vb * 3) + pix] = b0; out_color[(vb * 3) + hw + pix] = b1; out_color[(vb * 3) + 2 * hw + pix] = b2;
            if (aux.clamped) {
                aux.clamped[(vb * 3) + pix] = fminf(fmaxf(b0, 0.f), 1.f);
                aux.clamped[(vb * 3) + hw + pix] = fminf(fmaxf(b1, 0.f), 1.f);
                aux.clamped[(vb * 3) + 2 * hw + pix] = fminf(fmaxf(b2, 0.f), 1.f);
            }
            out_depth[vb + pix] = 0.f;
            out_alpha[vb + pix] = 0.f;
        }
        if constexpr (AUX >= 3) {
            // FUSED: the background pixels' share of the loss and of dL/dcolor (no Gaussian receives it); wave w's sum goes to the slot of quadrant w
            float lsum = 0.f;
            if (bx < W && by < H) {
                const size_t hw = (size_t)H * W, pix = (size_t)by * W + bx, vb = (size_t)view * hw;
                const float m = fz.mask ? fz.mask[vb + pix] : 1.f;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float x = bg[c], tg = fz.target[vb * 3 + (size_t)c * hw + pix];
                    const float d = (fminf(fmaxf(x, 0.f), 1.f) - tg) * m;
                    lsum += fabsf(d);
                    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                    fz.gimg[vb * 3 + (size_t)c * hw + pix] = (x >= 0.f && x <= 1.f) ? fz.weight * m * sg : 0.f;
                }
            }
            lsum = sgr_wave_sum(lsum);
            if (lane == 0) fz.loss_part[(size_t)bid * 4 + wave] = fz.weight * lsum;
        }
        return;
    }
    constexpr int kLossWave = 1;
    if constexpr (AUX >= 3) {
        // the loss needs the pixel's target colour and mask when the last survivor is composited, i.e. at the end of the workgroup's critical
        // path: the wave that will need them requests them NOW, straight into LDS (global_load_lds: no registers held, nobody waits until the epilogue)
        if (wave == kLossWave && inside) {                       // (the loss wave of the epilogue: the wave that waits for them)
            const size_t hw = (size_t)H * W, pix = (size_t)py * W + px, vb = (size_t)view * hw;
#pragma unroll
            for (int c = 0; c < 3; c++) __builtin_amdgcn_global_load_lds(fz.target + vb * 3 + (size_t)c * hw + pix, &sTgt[c][0], 4, 0, 0);
            if (fz.mask) __builtin_amdgcn_global_load_lds(fz.mask + vb + pix, &sTgt[3][0], 4, 0, 0);
        }
    }
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;      // this wave's share of the pixel sums
    float Tcarry = 1.f;                                         // identical in all 8 waves
    float cc0 = 0.f, cc1 = 0.f, cc2 = 0.f, ccD = 0.f, ccA = 0.f;   // composited-so-far (all waves), AUX only
    float Tstop = -1.f;
    uint32_t last = 0;
    uint32_t kbase = 0;                                         // survivors composited in earlier rounds
    uint32_t qhead = 0, qcount = 0;                             // LDS ring of survivors waiting to be composited
    size_t slot_next = (size_t)q * aux.NS + (range.x >> 6) + (size_t)bid;
    if (t < 64) sTstop[t] = -1.f;
    bool pix_done = !inside;
    // ---- software pipeline of the list: records of sub-chunk `commit` and ids of sub-chunk `commit + 1` live in registers
    int commit = 0;                                             // first list entry of the sub-chunk held in (ra, rb, rc, rid)
    float4 ra, rb, rc;
    uint32_t rid = 0, id_nx = 0;
    ra = rb = rc = make_float4(0.f, 0.f, 0.f, -1.f);
    if (t < n) {
        rid = point_list[range.x + t];
        ra = rec[(size_t)rid * 4 + 0]; rb = rec[(size_t)rid * 4 + 1]; rc = rec[(size_t)rid * 4 + 2];
    }
    if (t + kSegThreads < n) id_nx = point_list[range.x + t + kSegThreads];
    for (;;) {
        if (__syncthreads_count(pix_done) == kSegThreads) break;     // also: every wave is done reading the previous round's ring entries
        SGR_TR(2);
        // ---- fill: cull + append sub-chunks until a full round is available
        while (qcount < (uint32_t)(kSegWaves * kSegPer) && commit < n) {
            const int idx = commit + t;
            const bool bit = (idx < n) && cull_quadrant(ra, rc, qx0, qy0);
            const uint64_t bal = __ballot(bit);
            if (lane == 0) sWaveCnt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t woff = 0, m = 0;
#pragma unroll
            for (int w = 0; w < kSegWaves; w++) { const uint32_t cw = sWaveCnt[w]; if (w < wave) woff += cw; m += cw; }
            if (bit) {
                const uint32_t ord = qcount + woff + (uint32_t)__popcll(bal & lt_mask);     // position behind the ring head
                const uint32_t s = (qhead + ord) & (kSegRing - 1);
                const uint32_t pr = s >> 1, h = s & 1u;
                float *fa = (float *)&pA[pr], *fb = (float *)&pB[pr], *fc = (float *)&pC[pr], *fd = (float *)&pD[pr], *fe = (float *)&pE[pr];
                float *fq = (float *)&pQ[pr];
                fa[h] = ra.x; fa[2 + h] = ra.y;
                fb[h] = kHalfLog2e * ra.z; fb[2 + h] = kLog2e * ra.w;
                fc[h] = kHalfLog2e * rb.x; fc[2 + h] = rb.y;
                fd[2 * h] = rb.w; fd[2 * h + 1] = rc.x;
                fe[2 * h] = rc.y; fe[2 * h + 1] = rb.z;
                fq[h] = rc.w; fq[2 + h] = __uint_as_float((uint32_t)idx + 1u);       // rc.w = p*
                if (AUX && aux.ckpt_tc) aux.compact[(size_t)q * aux.R + range.x + kbase + ord] = make_uint2(rid, (uint32_t)idx);
            }
            qcount += m;
            commit += kSegThreads;
            // advance the pipeline: records of the next sub-chunk (its ids arrived a stage ago), ids of the one after
            const int nidx = commit + t;
            if (nidx < n) {
                rid = id_nx;
                ra = rec[(size_t)rid * 4 + 0]; rb = rec[(size_t)rid * 4 + 1]; rc = rec[(size_t)rid * 4 + 2];
            }
            if (nidx + kSegThreads < n) id_nx = point_list[range.x + nidx + kSegThreads];
            __syncthreads();                                     // ring entries visible; sWaveCnt reusable
        }
        if (qcount == 0) break;
        SGR_TR(3);
        // one round: m survivors in up to 8 segments of 16, 32 or 64 (1, 2 or 4 backward rows; segments never straddle a
        // 64-survivor bucket).  Only the last round of a list is shorter than 512, and then spreads over all waves.
        const uint32_t m = min(qcount, (uint32_t)(kSegWaves * kSegPer));
        const uint32_t per = m <= 16u * kSegWaves ? 16u : (m <= 32u * kSegWaves ? 32u : 64u);
        const uint32_t s0 = min(m, (uint32_t)wave * per), s1 = min(m, s0 + per);
        const uint32_t brow0 = (s0 & 63u) >> 4;                     // my segment's first row inside its 64-survivor bucket
        // the loops below take survivors four at a time without bounds checks: pad the (last) round with null entries (opacity 0).
        // qhead and every segment start are multiples of 4, so a group of four never wraps around the ring.
        if ((m & 3u) && (uint32_t)t < 4u - (m & 3u)) {
            const uint32_t s = (qhead + m + (uint32_t)t) & (kSegRing - 1);
            const uint32_t pr = s >> 1, h = s & 1u;
            float *fa = (float *)&pA[pr], *fb = (float *)&pB[pr], *fc = (float *)&pC[pr], *fd = (float *)&pD[pr], *fe = (float *)&pE[pr];
            float *fq = (float *)&pQ[pr];
            fa[h] = 0.f; fa[2 + h] = 0.f; fb[h] = 0.f; fb[2 + h] = 0.f; fc[h] = 0.f; fc[2 + h] = 0.f;      // opacity 0 ...
            fd[2 * h] = 0.f; fd[2 * h + 1] = 0.f; fe[2 * h] = 0.f; fe[2 * h + 1] = 0.f;
            fq[h] = kNever; fq[2 + h] = 0.f;                                                                  // ... and p* = +inf: never valid
        }
        if (m & 3u) __syncthreads();
#define SGR_RING(S) ((qhead + (S)) & (kSegRing - 1))
        // ---- phase 1: transmittance product of my segment
        float Tseg = 1.f;
        const v2f px2 = {pxf, pxf}, py2 = {pyf, pyf};
        for (uint32_t s = s0; s < s1; s += 4) {
            float om[4];
            const uint32_t pb = SGR_RING(s) >> 1;
#pragma unroll
            for (int u = 0; u < 2; u++) {                                  // two pairs = four survivors
                const float4 qa = pA[pb + u], qb = pB[pb + u], qc = pC[pb + u];
                const float2 ps = *reinterpret_cast<const float2 *>(&pQ[pb + u]);      // (p*0, p*1)
                const v2f gx = {qa.x, qa.y}, gy = {qa.z, qa.w}, kxx = {qb.x, qb.y}, kxy = {qb.z, qb.w}, kyy = {qc.x, qc.y}, op = {qc.z, qc.w};
                const v2f dx = gx - px2, dy = gy - py2;
                const v2f power = __builtin_elementwise_fma(dx, kxx * dx, __builtin_elementwise_fma(kxy * dx, dy, (kyy * dy) * dy));   // == sgr_power2 per element
                const v2f G = {__builtin_amdgcn_exp2f(power.x), __builtin_amdgcn_exp2f(power.y)};
                const v2f og = op * G;
                const float a0 = fminf(0.99f, og.x), a1 = fminf(0.99f, og.y);
                om[2 * u] = ((power.x <= 0.f) & (power.x >= ps.x)) ? 1.f - a0 : 1.f;          // p* <= power <= 0: the published alpha test
                om[2 * u + 1] = ((power.y <= 0.f) & (power.y >= ps.y)) ? 1.f - a1 : 1.f;
            }
            Tseg = (((Tseg * om[0]) * om[1]) * om[2]) * om[3];
        }
        SGR_TR(4);
        sT[wave][lane] = Tseg;
        __syncthreads();
        SGR_TR(5);
        float Tin = Tcarry, Tall = Tcarry;
#pragma unroll
        for (int w = 0; w < kSegWaves; w++) { const float tw = sT[w][lane]; if (w < wave) Tin *= tw; Tall *= tw; }
        // ---- phase 2: composite my segment from T_in with the published sequential rule.
        // The transmittance in front of survivor k is evaluated as T_in * P_k with P_k the running product of the segment's (1 - alpha) from 1.0 -- the
        // very sequence phase 1 multiplied -- not as a product chained on from T_in: then the T_in of every later segment, T_in * P_last of this one, is
        // (rounding is monotone, P never grows) at most the value this segment tested at its stop, so "a pixel that stopped in an earlier segment has
        // T_in < 1e-4" holds in floating point, not just on paper.  With the chained product the two could disagree when T crossed 1e-4 within
        // rounding: a later segment then composited on and stopped the pixel a second time, and final_T -- two waves writing one LDS word --
        // differed from run to run (found by tools/fuzz_fused_step.py: one pixel in ~100 random scenes).
        float T = Tin, P = 1.f;
        bool done = !inside | (Tin < 0.0001f);
        const bool done_at_start = done;
        v2f d01 = {0.f, 0.f}, d2D = {0.f, 0.f};
        float dA = 0.f;
        uint32_t contributed = 0;
        // (a segment behind the point where all 64 pixels have stopped skips the loop: the code below is branch-free, so it would cost
        // the full 51 instructions per survivor for nothing -- a third of the evaluated pairs on the opaque C2 subject)
        if (__ballot(!done))
        for (uint32_t s = s0; s < s1; s += 4) {
            if constexpr (AUX != 0) {
                if (s != s0 && ((s - s0) & 15u) == 0u) {
                    // state at survivor 16 / 32 / 48 of my segment, relative to the segment start (the start's absolute sums are only
                    // known after the cross-wave prefix below; the backward adds the two)
                    const size_t sl = ((slot_next + (s0 >> 6)) * 4 + brow0 + ((s - s0) >> 4)) * 64 + lane;
                    if (aux.ckpt_tc) aux.ckpt_tc[sl] = make_float4(T, d01.x, d01.y, d2D.x);
                    if (aux.ckpt_da) aux.ckpt_da[sl] = make_float2(d2D.y, dA);
                }
            }
            float al[4];
            v2f rg[4], bd[4];                                              // (r, g) and (b, depth) of the four survivors
            uint32_t li[4];
            const uint32_t pb = SGR_RING(s) >> 1;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const float4 qa = pA[pb + u], qb = pB[pb + u], qc = pC[pb + u], qd = pD[pb + u], qe = pE[pb + u];
                const float4 qq = pQ[pb + u];
                const v2f gx = {qa.x, qa.y}, gy = {qa.z, qa.w}, kxx = {qb.x, qb.y}, kxy = {qb.z, qb.w}, kyy = {qc.x, qc.y}, op = {qc.z, qc.w};
                const v2f dx = gx - px2, dy = gy - py2;
                const v2f power = __builtin_elementwise_fma(dx, kxx * dx, __builtin_elementwise_fma(kxy * dx, dy, (kyy * dy) * dy));   // == sgr_power2 per element
                const v2f G = {__builtin_amdgcn_exp2f(power.x), __builtin_amdgcn_exp2f(power.y)};
                const v2f og = op * G;
                const float a0 = fminf(0.99f, og.x), a1 = fminf(0.99f, og.y);
                al[2 * u] = ((power.x <= 0.f) & (power.x >= qq.x)) ? a0 : 0.f;
                al[2 * u + 1] = ((power.y <= 0.f) & (power.y >= qq.y)) ? a1 : 0.f;
                rg[2 * u] = (v2f){qd.x, qd.y}; rg[2 * u + 1] = (v2f){qd.z, qd.w};
                bd[2 * u] = (v2f){qe.x, qe.y}; bd[2 * u + 1] = (v2f){qe.z, qe.w};
                li[2 * u] = __float_as_uint(qq.z); li[2 * u + 1] = __float_as_uint(qq.w);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                P *= 1.f - al[u];
                const float test_T = Tin * P;
                done = done | (test_T < 0.0001f);                 // the crossing Gaussian is NOT composited
                const bool contrib = (al[u] > 0.f) & !done;
                const float w = contrib ? al[u] * T : 0.f;
                const v2f w2 = {w, w};
                d01 = rg[u] * w2 + d01;                           // (d0, d1) += (r, g) * w   -- one v_pk_fma_f32
                d2D = bd[u] * w2 + d2D;                           // (d2, dD) += (b, depth) * w
                dA += w;
                T = contrib ? test_T : T;
                last = contrib ? li[u] : last;
                contributed |= contrib ? 1u : 0u;
            }
        }
        const float d0 = d01.x, d1 = d01.y, d2 = d2D.x, dD = d2D.y;
#undef SGR_RING
        SGR_TR(6);
        if (done && !done_at_start) Tstop = T;                   // I am the segment in which this pixel stopped
        C0 += d0; C1 += d1; C2 += d2; D += dD; A += dA;
        if (AUX) {
            sAcc[wave][0][lane] = d0; sAcc[wave][1][lane] = d1; sAcc[wave][2][lane] = d2; sAcc[wave][3][lane] = dD; sAcc[wave][4][lane] = dA;
            const uint64_t any_contrib = __ballot(contributed != 0u);
            if (lane == 0) sContrib[wave] = any_contrib ? 1u : 0u;
            __syncthreads();
            float p0 = cc0, p1 = cc1, p2 = cc2, pD = ccD, pA = ccA;
#pragma unroll
            for (int w = 0; w < kSegWaves; w++) {
                const float e0 = sAcc[w][0][lane], e1 = sAcc[w][1][lane], e2 = sAcc[w][2][lane], e3 = sAcc[w][3][lane], e4 = sAcc[w][4][lane];
                if (w < wave) { p0 += e0; p1 += e1; p2 += e2; pD += e3; pA += e4; }
                cc0 += e0; cc1 += e1; cc2 += e2; ccD += e3; ccA += e4;
            }
            if (s0 < m) {
                // only buckets in which some pixel of the quadrant composited something can receive gradient
                uint32_t live = 0;                                    // any segment of my bucket (1, 2 or 4 waves share one)
#pragma unroll
                for (int w = 0; w < kSegWaves; w++) if ((((uint32_t)w * per) >> 6) == (s0 >> 6)) live |= sContrib[w];
                if (live) {
                    const size_t slot = slot_next + (s0 >> 6);
                    if (kbase + s0 != 0u) {
                        if (aux.ckpt_tc) aux.ckpt_tc[(slot * 4 + brow0) * 64 + lane] = make_float4(Tin, p0, p1, p2);
                        if (aux.ckpt_da) aux.ckpt_da[(slot * 4 + brow0) * 64 + lane] = make_float2(pD, pA);
                    }
                    if (lane == 0 && (s0 & 63u) == 0u && aux.ckpt_tc)
                        aux.desc[slot] = make_uint2(bid | (((per >> 4) - 1u) << 30), ((kbase + s0) << 7) | min(64u, m - s0));
                }
            }
            slot_next += (m + 63u) >> 6;
        }
        SGR_TR(7);
#ifdef SGR_SEG_TRACE
        if (sch) { sch[4] += 1ull; sch[5] = kbase + m; }          // rounds, survivors composited
#endif
        kbase += m;
        qhead = (qhead + m) & (kSegRing - 1);
        qcount -= m;
        Tcarry = Tall;
        pix_done = !inside | (Tcarry < 0.0001f);
    }
    SGR_TR(8);
    // ---- combine the 8 partial sums
    __syncthreads();
    if (Tstop >= 0.f) sTstop[lane] = Tstop;
    sAcc[wave][0][lane] = C0; sAcc[wave][1][lane] = C1; sAcc[wave][2][lane] = C2; sAcc[wave][3][lane] = D; sAcc[wave][4][lane] = A;
    sLast[wave][lane] = last;
    __syncthreads();
    // wave 0 writes the pixels' outputs; the fused step's loss share and dL/dcolor are wave 1's, side by side with wave 0's stores (the epilogue
    // is on every workgroup's critical path)
    if ((wave == 0 || (AUX >= 3 && wave == kLossWave)) && !(AUX && !aux.ckpt_tc)) {
        float lsum = 0.f;
        if (inside) {
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, rD = 0.f, rA = 0.f;
            uint32_t lmax = 0;
#pragma unroll
            for (int w = 0; w < kSegWaves; w++) {
                r0 += sAcc[w][0][lane]; r1 += sAcc[w][1][lane]; r2 += sAcc[w][2][lane]; rD += sAcc[w][3][lane]; rA += sAcc[w][4][lane];
                lmax = max(lmax, sLast[w][lane]);
            }
            const float ts = sTstop[lane];
            const float Tf = ts >= 0.f ? ts : Tcarry;
            const size_t hw = (size_t)H * W;
            const size_t pix = (size_t)py * W + px;
            const size_t vb = (size_t)view * hw;
            const float o0 = r0 + Tf * bg[0], o1 = r1 + Tf * bg[1], o2 = r2 + Tf * bg[2];
            float tgt[3] = {0.f, 0.f, 0.f}, m = 1.f;
            if (AUX >= 3 && wave == kLossWave) {
                // (the direct-to-LDS loads of the prologue landed long ago; the wait stands in FRONT of this wave's stores, or it would be a wait for
                // them as well)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tgt[0] = sTgt[0][lane]; tgt[1] = sTgt[1][lane]; tgt[2] = sTgt[2][lane];
                if (fz.mask) m = sTgt[3][lane];
            }
            if (wave == 0) {
                final_T[vb + pix] = Tf;
                n_contrib[vb + pix] = lmax;
                out_color[(vb * 3) + pix] = o0;
                out_color[(vb * 3) + hw + pix] = o1;
                out_color[(vb * 3) + 2 * hw + pix] = o2;
                if (aux.clamped) {
                    aux.clamped[(vb * 3) + pix] = fminf(fmaxf(o0, 0.f), 1.f);
                    aux.clamped[(vb * 3) + hw + pix] = fminf(fmaxf(o1, 0.f), 1.f);
                    aux.clamped[(vb * 3) + 2 * hw + pix] = fminf(fmaxf(o2, 0.f), 1.f);
                }
                out_depth[vb + pix] = rD;
                out_alpha[vb + pix] = rA;
            }
            if (AUX >= 3 && wave == kLossWave) {
                // the clamp + masked L1 of loss.hip (clamped_l1_kernel), per pixel the same expressions: dL/dcolor is bit-identical to that kernel's
                const float oc[3] = {o0, o1, o2};
                float g[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float tg = tgt[c];
                    const float d = (fminf(fmaxf(oc[c], 0.f), 1.f) - tg) * m;
                    lsum += fabsf(d);
                    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                    g[c] = (oc[c] >= 0.f && oc[c] <= 1.f) ? fz.weight * m * sg : 0.f;
                    fz.gimg[vb * 3 + (size_t)c * hw + pix] = g[c];
                }
            }
        }
        if (AUX >= 3 && wave == kLossWave) {
            lsum = sgr_wave_sum(lsum);
            if (lane == 0) fz.loss_part[(size_t)bid * 4 + q] = fz.weight * lsum;
        }
    }
#ifdef SGR_SEG_TRACE
    SGR_TR(9);
    if (tr_buf) tr_buf[0] = (unsigned long long)tr_n;
    if (sch) sch[1] = wall_clock64();
#endif
}

// -------------------------------------------------------------------------------------------------
// Work order for the segment-parallel forward (one workgroup): occupied tiles, longest lists first, so that the tiles on the
// critical path start at t = 0 and the tail of the launch is made of short ones (the unordered launch started the heaviest
// C2 tiles 35 us late behind 3000 empty workgroups).  32 length classes (n >> 7), order inside a class is arbitrary -- it only
// affects scheduling, never results.  Also clears the bucket descriptors (replaces a memset launch).
// order = uint4 per slot: (tile id, first, end of its list, 0), longest list first, empty tiles last.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fwd_prepare_kernel(const uint2 *__restrict__ ranges, uint32_t tiles_total, uint2 *__restrict__ desc,
                                                           size_t n_desc, uint32_t *__restrict__ order) {
    __shared__ uint32_t sTmp[66];
    sgr_fwd_prepare(ranges, tiles_total, desc, n_desc, order, sTmp);
}

// FUSED step: the per-(tile, quadrant) loss shares -> per-view sums and their total, added in a fixed order by ONE workgroup of NW waves (no
// atomics: unlike clamped_l1_kernel's float atomics the loss is bitwise reproducible).
struct LossReduce { const float *loss_part; float *loss_view, *loss_total; uint32_t per_view; int n_views; uint32_t block; };
template <int NW>
__device__ __forceinline__ void l1_reduce_block(const LossReduce &r) {
    __shared__ float red[NW];
    float total = 0.f;
    for (int v = 0; v < r.n_views; v++) {
        float acc = 0.f;
        for (uint32_t i = threadIdx.x; i < r.per_view; i += 64u * NW) acc += r.loss_part[(size_t)v * r.per_view + i];
        acc = sgr_wave_sum(acc);
        if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) sum += red[w];
            r.loss_view[v] = sum;
            total += sum;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && r.loss_total) *r.loss_total = total;
}

// -------------------------------------------------------------------------------------------------
// B1: bucket-parallel "systolic" backward.  One wave = one bucket of <= 64 consecutive surviving Gaussians of one (tile, quadrant)
// against that quadrant's pixels.  LANES OWN GAUSSIANS: each lane keeps its Gaussian's record and its ten gradient accumulators in
// registers for the whole kernel, so there is no cross-lane reduction; the pixel states (the running transmittance and the
// composited-so-far dot product) travel through the lanes of a 16-lane row by DPP shifts, front to back, starting from the forward
// pass's checkpoint for that row; the static pixel data is read from LDS.  All buckets of a frame run concurrently (thousands of
// independent waves instead of one serial walk per tile), and each (tile instance, quadrant) pair leaves ONE non-atomic 40-byte partial
// record that preprocess_bwd gathers in a fixed order: NO ATOMICS anywhere in the backward, gradients are bitwise reproducible.
//
// Math (front-to-back form of the published reverse walk), per pixel with g = upstream gradient vector over
// (r,g,b,depth,alpha), f_j = (r_j,g_j,b_j,depth_j,1), q_j = f_j . g, w_j = alpha_j T_j:
//   dL/dalpha_j = T_j q_j - (O - Pre_j - w_j q_j) / (1 - alpha_j),   O = out . g (includes the T_final*bg term),
//   Pre_j = sum_{k<j} w_k q_k  (running), T_{j+1} = T_j (1 - alpha_j)  (the wave forward's T sequence bit for bit; the segment-parallel forward
//   evaluates T_in * P_k, equal to a last bit or two: its phase 2).
// -------------------------------------------------------------------------------------------------
// shift one lane up inside each 16-lane row; lane 0 of every row takes `feed` (DPP row_shr:1 keeps `old` where there is no source)
__device__ __forceinline__ float row_shift_in(float v, float feed) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, feed), __builtin_bit_cast(int, v),
                                                                 SGR_DPP_ROW_SHR(1), 0xF, 0xF, false));
}


// The step of the bucket backward's pixel pipeline (render_bwd_bucket_kernel below explains it).  Names taken from the expansion site: gx gy kxx
// kyy kxy op pstar gidx cr cg cb gdep (my Gaussian), rl (lane & 15), n_alive, pa pb pd (this wave's LDS feeds), the ten sums S1 .. a9, HAS_DA.
#define SGR_BWD_LOAD(S, FA, FB, FD) { FA = pa[(S)]; FB = pb[(S)]; FD = pd[min((S), n_alive - 1)]; asm volatile("" ::: "memory"); }
#define SGR_BWD_STEP(IN, OUT, S, fa, fb, fd, NFA, NFB, NFD)                                                             \
    {                                                                                                                   \
        const bool has = (uint32_t)((S) - rl) < (uint32_t)n_alive;      /* a stream entry sits in this lane */           \
        /* the shift is the first use of this step's reads (one wait, nothing else in flight: the compiler's wait counts do not   \
           survive the loop's back edge, a wait with the next reads already issued would be a wait for them too); THEN the next   \
           step's reads go out and travel while this step computes */                                                             \
        OUT.T = row_shift_in(IN.T, fd.x); OUT.Rem = row_shift_in(IN.Rem, fd.y);                                         \
        SGR_BWD_LOAD((S) + 1, NFA, NFB, NFD)                                                                            \
        const float dx = gx - fa.x, dy = gy - fa.y;                                                                     \
        const float p2 = sgr_power2(kxx, kyy, kxy, dx, dy);                                                             \
        const float G = __builtin_amdgcn_exp2f(p2);                                                                     \
        const float alpha = fminf(0.99f, op * G);                                                                       \
        const bool valid = has && gidx < __float_as_uint(fa.z) && p2 <= 0.f && p2 >= pstar;                              \
        if (valid) {                                                                                                    \
            const float w = alpha * OUT.T;                                                                              \
            float qj = sgr_dot3(cr, fa.w, cg, fb.x, cb, fb.y);                                                          \
            if (HAS_DA) qj += fmaf(gdep, fb.z, fb.w);                                                                   \
            const float oma = 1.f - alpha;                                                                              \
            OUT.Rem = fmaf(-w, qj, OUT.Rem);                                                                            \
            const float dL_dalpha = fmaf(OUT.T, qj, -(OUT.Rem * __builtin_amdgcn_rcpf(oma)));                           \
            OUT.T *= oma;                                                                                               \
            const float v = G * dL_dalpha; /* upstream differentiates through op*G even when alpha is capped */         \
            const float vx = v * dx, vy = v * dy;                                                                       \
            S1 += v; Sx += vx; Sy += vy;                                                                                \
            Sxx = fmaf(vx, dx, Sxx); Sxy = fmaf(vx, dy, Sxy); Syy = fmaf(vy, dy, Syy);                                  \
            if (HAS_DA) aD = fmaf(w, fb.z, aD);                                                                         \
            a7 = fmaf(w, fa.w, a7); a8 = fmaf(w, fb.x, a8); a9 = fmaf(w, fb.y, a9);                                     \
        }                                                                                                               \
    }

// One wave = one bucket of <= 64 consecutive surviving Gaussians of a (tile, quadrant), run as FOUR independent 16-lane
// pipelines: row r owns survivors 16r..16r+15 and starts from the forward's checkpoint for that row.  The stream of the quadrant's
// pixels that can still receive something from the bucket enters lane 0 of every row, one pixel per step, and moves one lane up per
// step, so a pixel meets the row's Gaussians in front-to-back order: (stream length + 15) steps, at most 79, instead of the 127 a
// single 64-lane pipeline needs (fill/drain is 15 steps instead of 63).
// SPLIT (launches with few buckets, i.e. one or two views): TWO waves per bucket, each streaming one half of the quadrant's pixels
// -- more wave-steps but twice the waves, which is what a 4 000-bucket launch on 1 024 SIMDs lacks; the second wave's per-Gaussian
// sums are added to the first's through LDS before the single partial record is written.
template <bool HAS_DA, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 128 : 64) void render_bwd_bucket_kernel(int W, int H, int Tx, uint32_t tiles_per_view,
                                                                   const uint2 *__restrict__ ranges,
                                                                   const float4 *__restrict__ rec,
                                                                   const uint4 *__restrict__ rect,
                                                                   const uint32_t *__restrict__ n_contrib,
                                                                   const float *__restrict__ out_color,
                                                                   const float *__restrict__ out_depth,
                                                                   const float *__restrict__ out_alpha,
                                                                   const float *__restrict__ gC, const float *__restrict__ gD,
                                                                   const float *__restrict__ gA, const float *__restrict__ gscale,
                                                                   FwdAux aux, float4 *__restrict__ part, uint8_t *__restrict__ flags, int clamp_grad,
                                                                   LossReduce lred) {
    // (FUSED step: one extra workgroup behind the bucket slots adds up the loss shares the compositing kernel left: no launch of its own)
    if (lred.loss_part && blockIdx.x == lred.block) { l1_reduce_block<SPLIT ? 2 : 1>(lred); return; }
    // !SPLIT (batches of views): ONE wave = one bucket per workgroup.  Most bucket slots of a launch are unused (the slot count is an upper
    // bound from the list lengths: 78 % empty at C3); with four buckets per workgroup a workgroup usually held one real wave and 16 KB of
    // LDS until it was done, which capped a CU at ~10 working waves.
    constexpr int NWV = SPLIT ? 2 : 1;        // waves per workgroup (SPLIT: the two halves of one bucket)
    __shared__ float4 sPix[NWV][2][96];       // per wave: [0] (x, y, n_contrib bits, g0) of stream entry i at [16 + i] (16 unread-but-addressable
                                              //           [1] (g1, g2, gD, gA)                  slots on either side: lanes look 15 entries back and ahead)
    float4 (*sPixA)[96] = reinterpret_cast<float4 (*)[96]>(&sPix[0][0][0]);        // sPixA[w] = sPix[w][0]: rows 2 w, sPixB[w] = sPix[w][1]: rows 2 w + 1
    float4 (*sPixB)[96] = reinterpret_cast<float4 (*)[96]>(&sPix[0][1][0]);
    __shared__ float2 sDyn[NWV][4][64];       // per wave, per row: (T, Rem) of pixel p at the start of the row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCD placement as in the forward: workgroup ids b, b+8, b+16, b+24 (same XCD) take the same stretch of bucket slots in the four
    // quadrants, i.e. buckets of the same tiles, which gather the same records
    const uint32_t qb = (blockIdx.x >> 3) & 3u, idx = (blockIdx.x >> 5) * 8u + (blockIdx.x & 7u);
    constexpr int NPIX = SPLIT ? 32 : 64;                      // pixels streamed by one wave
    const uint32_t half = SPLIT ? (uint32_t)(wv & 1) : 0u;     // which half of the quadrant's pixels
    constexpr uint32_t sub = 0u;                                              // (one bucket per workgroup)
    if ((size_t)idx >= (size_t)aux.NS) return;
    const size_t slot = (size_t)qb * aux.NS + (size_t)idx;
    const uint2 desc_v = aux.desc[slot];
    // the descriptor is wave-uniform: move it to SGPRs so the step loop below is a scalar loop
    const uint32_t desc_y = (uint32_t)__builtin_amdgcn_readfirstlane((int)desc_v.y);
    const uint32_t desc_x = (uint32_t)__builtin_amdgcn_readfirstlane((int)desc_v.x);
    const uint32_t bid = desc_x & 0x3FFFFFFFu;
    const uint32_t rps = (desc_x >> 30) + 1u;                  // rows per forward segment; rows with r % rps == 0 hold absolute sums
    const uint32_t count = desc_y & 127u;
    if (count == 0) return;                                   // unused bucket slot (both waves of a SPLIT pair leave together; ended waves do not count at barriers)
    const uint32_t start = desc_y >> 7;                        // ordinal of this bucket's first survivor in the quadrant list
    const uint32_t q = qb;
    const uint32_t view = bid / tiles_per_view, tile = bid - view * tiles_per_view;
    const uint32_t tx = tile % Tx, ty = tile / Tx;
    const uint32_t rx = ranges[bid].x;
    // ---- my Gaussian: survivor 16*row + l of the bucket
    const int row = lane >> 4;
    const uint32_t gi = (uint32_t)lane;                        // == 16*row + (lane & 15)
    const bool has_g = gi < count;
    uint2 e = make_uint2(0u, 0xFFFFFFFFu);
    if (has_g) e = aux.compact[(size_t)q * aux.R + rx + start + gi];
    // a lane without a Gaussian gets list index 0xFFFFFFFF, which no pixel's n_contrib exceeds -> never valid
    const uint32_t gidx = e.y;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    uint4 rd = make_uint4(0u, 0u, 0u, 0u);                     // (rect min, rect max, depth bits, first tile-instance index)
    if (has_g) { ra = rec[(size_t)e.x * 4 + 0]; rb = rec[(size_t)e.x * 4 + 1]; rc = rec[(size_t)e.x * 4 + 2]; rd = rect[e.x]; }
    const float pstar = has_g ? rc.w : kNever;                 // alpha test: p* <= power <= 0 (file header)
    const float gx = ra.x, gy = ra.y, cxx = ra.z, cxy = ra.w, cyy = rb.x, op = rb.y, gdep = rb.z, cr = rb.w, cg = rc.x, cb = rc.y;
    // conic pre-scaled into the exp2 domain: G = exp(power) = exp2(kxx dx^2 + kyy dy^2 + kxy dx dy)
    const float kL2e = 1.4426950408889634f;
    const float kxx = -0.5f * kL2e * cxx, kyy = -0.5f * kL2e * cyy, kxy = -kL2e * cxy;
    // ---- pixel p = lane: static data and the four row start states go to LDS (the per-step feeders).  Only the pixels that can still
    // receive a contribution from this bucket are streamed: a pixel whose last contributor lies in front of the bucket's first survivor
    // (n_contrib <= its list index) is valid for none of the bucket's Gaussians.  The stream keeps the pixel order, so the sums of the
    // step loop add the same terms in the same order: bit-identical to streaming all 64, in (alive pixels + 15) steps instead of 79.
    const uint32_t gidx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gidx);        // list index of the bucket's first survivor (count >= 1)
    const int p = lane + (int)half * NPIX;
    const int px = (int)tx * 16 + (int)(q & 1u) * 8 + (p & 7);
    const int py = (int)ty * 16 + (int)(q >> 1) * 8 + (p >> 3);
    const bool inside = lane < NPIX && px < W && py < H;
    const size_t hw = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const size_t vb = (size_t)view * hw;
    uint32_t last = 0;
    if (inside) last = n_contrib[vb + pix];
    const bool alive = last > gidx0;
    const uint64_t alive_mask = __ballot(alive);
    const int n_alive = (int)__popcll(alive_mask);
    const int pos = (int)__popcll(alive_mask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));       // this pixel's place in the stream
    if (alive) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f, ga = 0.f, O = 0.f;
        {
            const float gs = gscale ? *gscale : 1.f;          // optional device scalar on dL/dcolor
            g0 = gs * gC[vb * 3 + pix]; g1 = gs * gC[vb * 3 + hw + pix]; g2 = gs * gC[vb * 3 + 2 * hw + pix];
            const float c0 = out_color[vb * 3 + pix], c1 = out_color[vb * 3 + hw + pix], c2 = out_color[vb * 3 + 2 * hw + pix];
            if (clamp_grad) {        // the upstream gradient is w.r.t. clamp(colour, 0, 1) (gs.py:107): torch.clamp's backward, inclusive mask
                g0 = (c0 >= 0.f && c0 <= 1.f) ? g0 : 0.f; g1 = (c1 >= 0.f && c1 <= 1.f) ? g1 : 0.f; g2 = (c2 >= 0.f && c2 <= 1.f) ? g2 : 0.f;
            }
            // O = out . g: everything the pixel composited (incl. the T_final*bg term), dotted with the upstream gradient
            O = sgr_dot3(c0, g0, c1, g1, c2, g2);
            if (HAS_DA) {
                if (gD) gd = gD[vb + pix];
                if (gA) ga = gA[vb + pix];
                O += fmaf(out_alpha[vb + pix], ga, out_depth[vb + pix] * gd);
            }
        }
        sPixA[2 * wv][16 + pos] = make_float4((float)px, (float)py, __uint_as_float(last), g0);
        sPixB[2 * wv][16 + pos] = make_float4(g1, g2, gd, ga);
        float T0 = 1.f, Pre0 = 0.f;
        if (start) {
            const float4 tc = aux.ckpt_tc[slot * 256 + p];
            T0 = tc.x;
            Pre0 = sgr_dot3(tc.y, g0, tc.z, g1, tc.w, g2);
            if (HAS_DA) { const float2 da = aux.ckpt_da[slot * 256 + p]; Pre0 += fmaf(da.y, ga, da.x * gd); }
        }
        sDyn[wv][0][pos] = make_float2(T0, O - Pre0);
        {
            float PreSeg = Pre0;                               // composited-so-far at the start of the forward segment the row is in
#pragma unroll
            for (int r = 1; r < 4; r++) {
                float Tr = 1.f, Prer = Pre0;
                if ((uint32_t)(16 * r) < count) {
                    const float4 tc = aux.ckpt_tc[(slot * 4 + r) * 64 + p];
                    Tr = tc.x;
                    float dotv = sgr_dot3(tc.y, g0, tc.z, g1, tc.w, g2);
                    if (HAS_DA) { const float2 da = aux.ckpt_da[(slot * 4 + r) * 64 + p]; dotv += fmaf(da.y, ga, da.x * gd); }
                    if (((uint32_t)r & (rps - 1u)) == 0u) PreSeg = dotv;      // rps is 1, 2 or 4
                    Prer = (((uint32_t)r & (rps - 1u)) == 0u) ? dotv : PreSeg + dotv;
                }
                sDyn[wv][r][pos] = make_float2(Tr, O - Prer);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- the pixel stream moves through the 16 lanes of every row, one lane per step: at step S lane l of a row holds stream entry S - l.
    // Only the pixel's DYNAMIC state (T, Rem) travels from lane to lane (v_mov_b32_dpp row_shr:1, two register sets that alternate
    // roles every step: a DPP shift-in writes its result over the fresh feed, so a single set would cost one copy per register and
    // step); its static data (position, n_contrib, upstream gradient) every lane reads for itself from LDS at [S - l]: consecutive
    // addresses across the lanes, the same LDS cycles as the broadcast reads they replace, six DPP moves per step fewer.
    struct PixState { float T, Rem; };
    PixState A = {1.f, 0.f}, B = A;
    // per-Gaussian moment accumulators of v = G * dL/dalpha over the pixels (constant factors applied once at the end)
    float S1 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, aD = 0.f, a7 = 0.f, a8 = 0.f, a9 = 0.f;
    const int nsteps = n_alive ? n_alive + (int)min(count, 16u) - 1 : 0;
    const int rl = lane & 15;
    // The three LDS reads of a step are issued one step AHEAD (two register sets) and pinned there by a compiler barrier: left to itself
    // the compiler sinks them into the regions that use them -- feed -> wait -> shift, position -> wait -> Gaussian, gradient -> wait ->
    // sums: three exposed LDS round trips per step on a kernel whose waves spend most of their time waiting.  (Volatile loads are no
    // alternative: each is followed by a wait for its completion.)
    const float4 *pa = &sPixA[2 * wv][16 - rl], *pb = &sPixB[2 * wv][16 - rl];          // [S] = stream entry S - rl
    const float2 *pd = &sDyn[wv][row][0];
    int s = 0;
    float4 fa0, fb0, fa1, fb1;
    float2 fd0, fd1;
    SGR_BWD_LOAD(0, fa0, fb0, fd0)               // (entries beyond the stream are addressable and unused: has == false)
    for (; s + 1 < nsteps; s += 2) {
        SGR_BWD_STEP(A, B, s, fa0, fb0, fd0, fa1, fb1, fd1)
        SGR_BWD_STEP(B, A, s + 1, fa1, fb1, fd1, fa0, fb0, fd0)
    }
    if (s < nsteps) SGR_BWD_STEP(A, B, s, fa0, fb0, fd0, fa1, fb1, fd1)
    if (SPLIT) {
        // the odd wave hands its sums to the even wave of the same bucket
        // (the odd wave's own pixel arrays, 3 KB, are free once ITS step loop is over: its ten sums go there -- 10 KB of LDS per workgroup instead of
        // 12.5: 16 workgroups = eight waves per SIMD on a CU instead of 12 = six, and a one-view launch's 8 200 waves are resident at once)
        float (*sComb)[10][64] = reinterpret_cast<float (*)[10][64]>(&sPix[NWV - 1][0][0]);
        static_assert(sizeof(float) * 10 * 64 <= sizeof(float4) * 2 * 96, "the odd wave's pixel arrays hold its ten sums");
        if (half == 1u) {
            float *c = &sComb[sub][0][lane];
            c[0] = S1; c[64] = Sx; c[128] = Sy; c[192] = Sxx; c[256] = Sxy; c[320] = Syy; c[384] = aD; c[448] = a7; c[512] = a8; c[576] = a9;
        }
        __syncthreads();
        if (half == 1u) return;
        const float *c = &sComb[sub][0][lane];
        S1 += c[0]; Sx += c[64]; Sy += c[128]; Sxx += c[192]; Sxy += c[256]; Syy += c[320]; aD += c[384]; a7 += c[448]; a8 += c[512]; a9 += c[576];
    }
    // a Gaussian that was valid for no pixel of the quadrant (all sums exactly zero) leaves no record: its flag stays clear and the gather
    // skips it -- adding its zeros would change nothing
    const bool nonzero = (S1 != 0.f) | (Sx != 0.f) | (Sy != 0.f) | (Sxx != 0.f) | (Sxy != 0.f) | (Syy != 0.f) | (aD != 0.f) | (a7 != 0.f) |
                         (a8 != 0.f) | (a9 != 0.f);
    // The flag of EVERY survivor this launch looked at is rewritten (1 = record present, 0 = none): the flags live in the forward's image
    // blob and are cleared only by the forward chain, so a second backward on the same forward state (retain_graph, autograd.grad twice
    // with another upstream gradient -- the reference's calculate_adaptive_weight does that) must not see the first one's flags next to
    // its own freshly allocated, partly unwritten record buffer.  (Instance, quadrant) pairs that are in no live bucket are never set.
    if (has_g) {
        const uint32_t off = rd.w, rmin = rd.x, rmax = rd.y;
        const uint32_t inst = off + (ty - (rmin >> 16)) * ((rmax & 0xFFFFu) - (rmin & 0xFFFFu)) + (tx - (rmin & 0xFFFFu));
        flags[(size_t)inst * 4 + q] = nonzero ? 1 : 0;
    }
    if (has_g && nonzero) {
        // one NON-atomic 40-byte partial record per (tile instance, quadrant); preprocess_bwd gathers them in a fixed order
        const uint32_t off = rd.w, rmin = rd.x, rmax = rd.y;
        const uint32_t inst = off + (ty - (rmin >> 16)) * ((rmax & 0xFFFFu) - (rmin & 0xFFFFu)) + (tx - (rmin & 0xFFFFu));
        const float a0 = -0.5f * (float)W * op * fmaf(cxy, Sy, cxx * Sx);       // dL/dNDC x (includes 0.5*W like upstream)
        const float a1 = -0.5f * (float)H * op * fmaf(cxy, Sx, cyy * Sy);
        // 40 B, 8-byte aligned: 16 + 16 + 8-byte stores (three write requests per record instead of five)
        struct __attribute__((packed, aligned(8))) Rec40 { float2 v[5]; };
        Rec40 rr;
        rr.v[0] = make_float2(a0, a1);
        rr.v[1] = make_float2(-0.5f * op * Sxx, -0.5f * op * Sxy);
        rr.v[2] = make_float2(-0.5f * op * Syy, S1);
        rr.v[3] = make_float2(aD, a7);
        rr.v[4] = make_float2(a8, a9);
        *(reinterpret_cast<Rec40 *>(part) + ((size_t)inst * 4 + q)) = rr;
    }
}

}  // namespace

int sgr_validate_problem(const SgrProblem *pb);
int sgr_render_forward_kind(const SgrProblem *pb);

// 0 = automatic (segment-parallel for <= 2048 tiles, else one wave per quadrant), 2 = segment-parallel kernel, 3 = one wave per
// (tile, quadrant) (dev/test override: sgr_set_forward_mode).  (1 was round 1's serial per-tile kernel: removed in round 5.)
static int sgr_fwd_mode_from_env() { const int v = sgr_env_knob("SIGMAN_FWD_MODE", 0, 3, 0); return v == 1 ? 0 : v; }       // (1 was removed in round 5: as refused as by the setter)
static thread_local int sgr_fwd_mode = sgr_fwd_mode_from_env();          // (dev/test switch, thread-local like sgr_set_debug: the forward runs on the caller's thread; every thread starts from the environment)
extern "C" int sgr_set_forward_mode(int mode) {
    if (mode != 0 && mode != 2 && mode != 3) { sgr_set_error("sgr_set_forward_mode: %d is not a forward kernel (0 automatic, 2 segment-parallel, 3 one wave per quadrant)", mode); return 1; }
    sgr_fwd_mode = mode;
    return 0;
}
int sgr_get_forward_mode() { return sgr_fwd_mode; }

// a (tile, quadrant) list of n entries has <= n survivors in <= floor(n / 64) + 1 buckets, and floor(a/64) + floor(n/64) <= floor((a+n)/64):
// slot base (range.x >> 6) + tile id leaves exactly that room
extern "C" uint64_t sgr_bucket_slots(uint64_t R, uint64_t tiles_total) { return (R >> 6) + tiles_total + 1; }

static FwdAux make_aux(void *compact, void *ckpt_tc, void *ckpt_da, void *desc, uint64_t R, uint64_t tiles_total, float *clamped = nullptr) {
    FwdAux a;
    a.compact = (uint2 *)compact; a.ckpt_tc = (float4 *)ckpt_tc; a.ckpt_da = (float2 *)ckpt_da; a.desc = (uint2 *)desc;
    a.R = (uint32_t)R; a.NS = (uint32_t)sgr_bucket_slots(R, tiles_total);
    a.clamped = clamped;
    return a;
}

static bool fwd_is_seg(uint64_t tiles_total) { return sgr_fwd_mode == 2 || (sgr_fwd_mode != 3 && tiles_total <= 2048); }

// the compositing kernel a forward of this problem uses on the calling thread: 2 = segment-parallel, 3 = one wave per quadrant
int sgr_render_forward_kind(const SgrProblem *pb) {
    const uint64_t tiles_total = (uint64_t)((pb->W + SGR_TILE - 1) / SGR_TILE) * ((pb->H + SGR_TILE - 1) / SGR_TILE) * pb->n_views;
    return fwd_is_seg(tiles_total) ? 2 : 3;
}

// does this launch want the one-workgroup prepare step (tile order + descriptor clear), and how many descriptors are there?
int sgr_render_forward_wants_prepare(const SgrProblem *pb, uint64_t R, bool use_aux, size_t *n_desc_out) {
    const uint64_t tiles_total = (uint64_t)((pb->W + SGR_TILE - 1) / SGR_TILE) * ((pb->H + SGR_TILE - 1) / SGR_TILE) * pb->n_views;
    const size_t n_desc = use_aux ? (size_t)4 * sgr_bucket_slots(R, tiles_total) : 0;
    if (n_desc_out) *n_desc_out = n_desc;
    return fwd_is_seg(tiles_total) ? 1 : 0;
}

int sgr_render_forward_ex(const SgrProblem *pb, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                          float *out_color, float *out_depth, float *out_alpha, float *final_T, uint32_t *n_contrib,
                          uint64_t R, void *aux_compact, void *aux_ckpt_tc, void *aux_ckpt_da, void *aux_desc,
                          uint32_t *aux_order, int prepared /* bits: 1 the work order is there (plain form), 2 it is there in the single-view path's
                          class-major form (the empty tiles' outputs are written, too), 4 the bucket descriptors are cleared */, int kind /* 0: choose (sgr_render_forward_kind); else the compositing kernel to
                          use: the depth/alpha checkpoint pass must repeat its forward's */, const SgrFusedL1Args *fused /* NULL, or: the
                          single-view fused step -- loss shares and dL/dcolor written by the segment-parallel kernel */,
                          bool bg_done /* the empty tiles' outputs (and loss shares) are already written: SgrBgJob */, void *stream_) {
    if (sgr_validate_problem(pb)) return 1;
    if (kind == 0) kind = sgr_render_forward_kind(pb);
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint32_t tiles = (uint32_t)Tx * Ty;
    hipStream_t stream = (hipStream_t)stream_;
    // aux_ckpt_da == NULL: no depth/alpha checkpoints; aux_ckpt_tc == NULL (and aux_ckpt_da given): the pass that adds them later
    const bool use_aux = aux_compact && (aux_ckpt_tc || aux_ckpt_da) && aux_desc;
    const bool da_pass = use_aux && !aux_ckpt_tc;
    FwdAux aux = make_aux(aux_compact, aux_ckpt_tc, aux_ckpt_da, aux_desc, R, (uint64_t)tiles * pb->n_views, da_pass ? nullptr : pb->color_clamped);
    const uint64_t tiles_total = (uint64_t)tiles * pb->n_views;
    if (use_aux && tiles_total >= (1ull << 30)) { sgr_set_error("too many tiles (%llu) for the bucket descriptors", (unsigned long long)tiles_total); return 1; }
    // few workgroups (one or two 512^2 views): trade 1.5x arithmetic for an 8x shorter dependency chain
    const bool seg = kind == 2;
    const size_t n_desc = use_aux ? (size_t)4 * aux.NS : 0;
    // prepared: bit 0 = the work order is there in the plain form, bit 1 = in the class-major form (the empty tiles' outputs are written, too),
    // bit 2 = the bucket descriptors are cleared
    const bool order_done = (prepared & 3) != 0, desc_done = (prepared & 4) != 0;
    const bool prep = seg && !order_done && (aux_order || (use_aux && !desc_done && n_desc <= (1u << 17)));     // one workgroup orders the tiles (and clears the descriptors)
    const bool prep_desc = prep && use_aux && !desc_done && n_desc <= (1u << 17);
    if (da_pass && !prepared) { sgr_set_error("sgr_render_forward: the depth/alpha checkpoint pass must run on a prepared forward"); return 1; }
    if (fused && !(seg && use_aux && aux_ckpt_tc && !aux_ckpt_da)) { sgr_set_error("sgr_render_forward: the fused step needs the segment-parallel kernel with row checkpoints and no depth/alpha checkpoints"); return 1; }
    if (use_aux && !desc_done && !prep_desc && !da_pass) SGR_CHECK_HIP(hipMemsetAsync(aux_desc, 0, n_desc * sizeof(uint2), stream));
    SgrProfScope _p(SGR_K_RENDER_FWD, stream);
    if (prep) {
        hipLaunchKernelGGL(fwd_prepare_kernel, dim3(1), dim3(1024), 0, stream, (const uint2 *)ranges, (uint32_t)tiles_total,
                           prep_desc ? (uint2 *)aux_desc : (uint2 *)nullptr, n_desc, aux_order);
        SGR_CHECK_LAUNCH("fwd_prepare_kernel");
    }
    if (seg) {
        const uint32_t seg_grid = (uint32_t)((tiles_total + 7) / 8) * 32u;       // 8 tile slots x 4 quadrants per group of 32 ids
#define SGR_LAUNCH_SEG(A)                                                                                                   \
        hipLaunchKernelGGL(render_fwd_seg_kernel<A>, dim3(seg_grid), dim3(kSegThreads), 0, stream, pb->W, pb->H, Tx, tiles,        \
                           (const uint2 *)ranges, point_list, (const float4 *)rec, pb->bg, out_color, out_depth, out_alpha, final_T,  \
                           n_contrib, aux, (const uint32_t *)aux_order, (uint32_t)tiles_total, fz, (bg_done || (prepared & 2)) ? 1 : 0, (prepared & 2) ? 1 : 0)
        FusedL1 fz;
        memset(&fz, 0, sizeof(fz));
        if (fused) {
            fz.target = fused->target; fz.mask = fused->mask; fz.weight = fused->weight; fz.gimg = fused->gimg; fz.loss_part = fused->loss_part;
        }
        if (!use_aux) SGR_LAUNCH_SEG(0); else if (!fused) SGR_LAUNCH_SEG(2); else SGR_LAUNCH_SEG(3);
#undef SGR_LAUNCH_SEG
        SGR_CHECK_LAUNCH("render_fwd_seg_kernel");
        return 0;                      // (fused: the bucket backward queued behind this launch sums the loss shares on the side)
    }
    const uint32_t wgrid = (uint32_t)((tiles_total + 7) / 8) * 32u;          // 8 tiles x 4 quadrants per group of 32 ids
#define SGR_LAUNCH_WAVE(A)                                                                                                  \
    hipLaunchKernelGGL(render_fwd_wave_kernel<A>, dim3(wgrid), dim3(64), 0, stream, pb->W, pb->H, Tx, tiles, (uint32_t)tiles_total,   \
                       (const uint2 *)ranges, point_list, (const float4 *)rec, pb->bg, out_color, out_depth, out_alpha, final_T,  \
                       n_contrib, aux)
    if (!use_aux) SGR_LAUNCH_WAVE(0); else SGR_LAUNCH_WAVE(2);
#undef SGR_LAUNCH_WAVE
    SGR_CHECK_LAUNCH("render_fwd_wave_kernel");
    return 0;
}

extern "C" int sgr_render_forward(const SgrProblem *pb, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                                  float *out_color, float *out_depth, float *out_alpha, float *final_T, uint32_t *n_contrib,
                                  uint64_t R, void *aux_compact, void *aux_ckpt_tc, void *aux_ckpt_da, void *aux_desc,
                                  uint32_t *aux_order, void *stream_) {
    return sgr_render_forward_ex(pb, ranges, point_list, rec, out_color, out_depth, out_alpha, final_T, n_contrib, R, aux_compact,
                                 aux_ckpt_tc, aux_ckpt_da, aux_desc, aux_order, false, 0, nullptr, false, stream_);
}

int sgr_render_backward_ex(const SgrProblem *pb, const uint32_t *ranges, const float *rec, const uint32_t *rect,
                           const uint32_t *n_contrib, const float *out_color, const float *out_depth, const float *out_alpha,
                           const float *grad_color, const float *grad_depth, const float *grad_alpha, const float *grad_color_scale, uint64_t R,
                           const void *aux_compact, const void *aux_ckpt_tc, const void *aux_ckpt_da, const void *aux_desc,
                           float *part, uint32_t *flags, bool flags_cleared, const SgrFusedL1Args *loss_reduce /* NULL, or: sum its loss shares on the side */,
                           void *stream_) {
    if (sgr_validate_problem(pb)) return 1;
    hipStream_t stream = (hipStream_t)stream_;
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint32_t tiles = (uint32_t)Tx * Ty;
    if (!(aux_compact && aux_ckpt_tc && aux_desc && out_color && out_depth && out_alpha && part && flags && rect && n_contrib && grad_color)) {
        sgr_set_error("sgr_render_backward: needs the forward's auxiliary outputs (compact lists, checkpoints, descriptors), its output images, rect, part and flags");
        return 1;
    }
    if ((grad_depth || grad_alpha) && !aux_ckpt_da) { sgr_set_error("sgr_render_backward: dL/ddepth or dL/dalpha given but the forward left no depth/alpha checkpoints"); return 1; }
    if (R > 0 && !flags_cleared) SGR_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)R * 4, stream));   // (else: cleared by the forward chain)
    SgrProfScope _p(SGR_K_RENDER_BWD, stream);
    FwdAux aux = make_aux((void *)aux_compact, (void *)aux_ckpt_tc, (void *)aux_ckpt_da, (void *)aux_desc, R, (uint64_t)tiles * pb->n_views);
    // few buckets (one or two views): two waves per bucket (SPLIT) to give the SIMDs enough waves to hide latencies
    const bool split = (uint64_t)tiles * pb->n_views <= 2048;
    const uint32_t nblocks = (aux.NS + 7u) / 8u * 32u;               // one workgroup per (bucket slot, quadrant), in groups of 8 slots x 4 quadrants
    LossReduce lred;
    memset(&lred, 0, sizeof(lred));
    if (loss_reduce) { lred.loss_part = loss_reduce->loss_part; lred.loss_view = loss_reduce->loss_per_view; lred.loss_total = loss_reduce->loss_total;
                       lred.per_view = tiles * 4u; lred.n_views = pb->n_views; lred.block = nblocks; }
#define SGR_LAUNCH_BWD(DA, SP)                                                                                              \
    hipLaunchKernelGGL((render_bwd_bucket_kernel<DA, SP>), dim3(nblocks + (loss_reduce ? 1u : 0u)), dim3(SP ? 128 : 64), 0, stream, pb->W, pb->H, Tx, tiles, \
                       (const uint2 *)ranges, (const float4 *)rec, (const uint4 *)rect, n_contrib, out_color, out_depth, out_alpha, grad_color, \
                       grad_depth, grad_alpha, grad_color_scale, aux, (float4 *)part, (uint8_t *)flags, pb->clamp_grad, lred)
    const bool da = grad_depth || grad_alpha;
    if (split && da) SGR_LAUNCH_BWD(true, true); else if (split) SGR_LAUNCH_BWD(false, true);
    else if (da) SGR_LAUNCH_BWD(true, false); else SGR_LAUNCH_BWD(false, false);
#undef SGR_LAUNCH_BWD
    SGR_CHECK_LAUNCH("render_bwd_bucket_kernel");
    return 0;
}

extern "C" int sgr_render_backward(const SgrProblem *pb, const uint32_t *ranges, const float *rec, const uint32_t *rect,
                                   const uint32_t *n_contrib, const float *out_color, const float *out_depth, const float *out_alpha,
                                   const float *grad_color, const float *grad_depth, const float *grad_alpha, const float *grad_color_scale,
                                   uint64_t R, const void *aux_compact, const void *aux_ckpt_tc, const void *aux_ckpt_da, const void *aux_desc,
                                   float *part, uint32_t *flags, void *stream_) {
    return sgr_render_backward_ex(pb, ranges, rec, rect, n_contrib, out_color, out_depth, out_alpha, grad_color, grad_depth, grad_alpha,
                                  grad_color_scale, R, aux_compact, aux_ckpt_tc, aux_ckpt_da, aux_desc, part, flags, false, nullptr, stream_);
}
