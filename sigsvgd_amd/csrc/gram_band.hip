// Signature-kernel Gram forward/backward for short paths whose REFINED grid has 129 .. 256 cells per side -- the
// reference's call shapes examples/script_sequential_distribution.ipynb (10 points, dyadic order 4: 144 cells) and
// examples/script_control_particle_maze.py:43-44 (30 points, order 3: 232 cells) -- and, for small launches, 65 .. 128 cells
// (BASELINE C1; examples/script_planning_obstacle_field.py:156-158,325).
//
// Same frame as gram_dyad.hip: everything around the sweeps on the COARSE grid (static kernel T x T in fp64, increment table
// D_coarse / (r^2 sqrt(12)) in LDS, block sums of S = K_fwd * U in fp64 LDS, 4-corner scatter and both contractions once per
// pair in fp32 on the differences x_m - y_n), gradient partial sums through the segment / item slabs of grad_reduce_kernel (no
// atomics between wavefronts, bit-reproducible).  The sweeps run over BANDS of 64 cell rows and all P columns, in the fp32
// difference form of gram_fast.hip (V = K[p+1][q] - K[p][q] carried along the row, one full-magnitude add per cell that never
// feeds back); a band hands its last row to the next through LDS, and the forward solution goes to a per-pair scratch in
// [band][step][lane] order (coalesced 256-B rows, read back through an eight-deep register ring).  ONE kernel, two schedules:
//   * band-parallel (the reference's sizes: few pairs): the bands of a pair on as many wavefronts of one workgroup, pipelined;
//   * serial (launches with many pairs): a wavefront per pair, its bands one after the other, several pairs per workgroup
//     around one staged column trajectory.
// Rounds 3-4 had a separate one-wavefront-per-pair kernel with compiler-scheduled steps (31 + 42 instructions per step against
// about 20 + 28 here, overhead included); it is gone.
//
// Reference semantics: sigkernel _SigKernelGram.forward/backward [RECALLED, SURVEY.md App. A]; static kernel
// src/kernels/_traj_kernels.py:176-195; callers src/inference/score.py:68-69.
#include "sig_common.h"

#include <atomic>

namespace sigsvgd {

struct BandArgs {
    const void *X, *Y, *go;
    void *K;
    double *rseg; // [owned tiles + workgroups][rows per tile][T*d]
    float *cslab; // [items][T*d] (symmetric launches)
    float *wsk;   // [gridDim.x][pairs per workgroup][bands][steps][64]: forward solution of the pair in work (gradient launches)
    size_t wsk_per_wave;
    unsigned char *kflag; // [A][B]: 1 where the fp32 solution of the pair cancelled (max |K_grid| > r max(|K|, 0.1), r = 2 / 4 / 8) or is ill-conditioned (d <= 3), as in
                          // gram_quad.hip): the launcher lets the coverage kernel solve those pairs' K again in fp64
    int io64, A, B, T, d, n, symw;
    int comprev; // the reverse sweep's full-magnitude add in two floats as well (see band_comp)
    TileMap tm;
    long long nitems;
    double inv_h;
#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long *stamps; // diagnostic build: wave-cycles per phase of the band-parallel kernel
#endif
};

// Diagnostic build (-DSIGSVGD_PHASE_STAMPS, scripts/dev/phase_stamps.py): s_memtime around the phases of the band-parallel kernel
#ifdef SIGSVGD_PHASE_STAMPS
#define SIGB_STAMP(i)                                                        \
    {                                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        ph_[i] += now_ - tlast_;                                             \
        tlast_ = now_;                                                       \
    }
#else
#define SIGB_STAMP(i)
#endif

namespace {
constexpr int BTMAX = 33;  // coarse points per path
constexpr int BPMAX = 256; // refined cells per side
constexpr int BPAD = 64;   // boundary rows: entry e lives at [BPAD + e]; lanes outside the grid write into the padding
__device__ __forceinline__ double b_ldany(const void *b, size_t i, int io64)
{
    return io64 ? static_cast<const double *>(b)[i] : (double)static_cast<const float *>(b)[i];
}
__device__ __forceinline__ void b_stany(void *b, size_t i, double v, int io64)
{
    if (io64)
        static_cast<double *>(b)[i] = v;
    else
        static_cast<float *>(b)[i] = (float)v;
}
} // namespace

namespace {
// ---- the band kernel ---------------------------------------------------------------------------------------------------
// Band-parallel schedule.  The reference's refined call shapes come with few pairs (notebook 5,050, maze 630): one wavefront
// per pair leaves most of the chip's 1,024 SIMDs with one wavefront or none, each walking nb (P + 63) dependent steps per
// sweep.  Here the nb bands of a pair run on nb wavefronts of ONE workgroup, as a pipeline: band b needs, on its step s, the
// entry lane 63 of band b - 1 wrote on ITS step s + 63, so it runs BLAG phases of BGS steps behind its neighbour, with a
// workgroup barrier between the phases (every wavefront of the workgroup runs the same phase schedule; a band outside its step
// range just waits).  A sweep takes ceil((P + 63) / BGS) + BLAG (nb - 1) phases instead of nb (P + 63) steps: 368 against 621
// step times at the notebook's shape, 544 against 1,180 at the maze script's.  A workgroup is one pair: nb wavefronts.
// Serial schedule (template parameter SER): see the kernel.  The arithmetic of every cell and the order of every sum (a coarse
// cell's fine rows lie inside one band, since r divides 64) do not depend on the schedule: K is the same bit for bit.
constexpr int BGS = 16; // steps per phase
constexpr int BLAG = 5; // phases a band runs behind its neighbour: BLAG * BGS > 62 + BGS + 1 steps (see the refills in the kernel)
constexpr int BPP = 1;  // pairs per workgroup of the band-parallel schedule (2: measured, slower -- twice the rounds at the reference's sizes)
constexpr int BUO = 96; // offset of entry 0 in a reverse boundary row (a phase reads down to entry P - 16 - sp0 >= -78)
// The increment table carries zeros on either side of every row and one row of zeros behind the last: a lane outside the
// grid (column < 0 or >= P on the ramps of its band, or a row >= P of the last band) reads gamma = 0, for which a step leaves
// V alone and copies the neighbour's value -- ahead of its row that is the boundary value 1 all the way down the lanes, so a
// lane starts its row from the right state with no activity test and none of the three selects in the step.  Columns reach
// -78 .. P + 78: band_zpad(r) coarse cells past a row's end.
__host__ __device__ inline int band_zpad(int r) { return 80 / r + 2; }

struct BandPLds {
    int yd, yf, yref;                        // shared by the workgroup
    int Sc, Dc, hK, hU, rowacc, misc, dump;  // inside a pair's block (dump: a block of 96 floats per wavefront of the pair)
    int hn;                                  // floats per boundary row
    int pair0, per_pair, total;
};
// (serial_slots: 0 = band-parallel, one pair per workgroup with nb - 1 boundary rows each way; n > 0 = serial, n pairs per
//  workgroup with ONE boundary row each way, reused in place)
__host__ __device__ inline BandPLds bandp_lds(int T, int P, int dpad, int serial_slots)
{
    auto up16 = [](int b) { return (b + 15) & ~15; };
    const int Tm = T - 1, cells = Tm * Tm, rows = T * dpad, nb = (P + 63) >> 6;
    const int hrows = serial_slots ? 1 : (nb > 1 ? nb - 1 : 1), ndump = serial_slots ? 1 : nb;
    BandPLds L;
    L.hn = 2 * BPAD + 64 * ((P + 62) / 64) + 80;
    int o = 0;
    L.yd = o;   o += up16(T * (dpad + 1) * 8);
    L.yref = o; o += up16(dpad * 8);
    L.yf = o;   o += up16(rows * 4);
    L.pair0 = o;
    int w = 0;
    L.Sc = w;     w += up16(cells * 8);
    L.misc = w;   w += 64;
    const int dtab = (Tm + 1) * (Tm + 2 * band_zpad(P / Tm)); // padded increment table (see band_zpad)
    L.Dc = w;     w += up16((dtab > rows ? dtab : rows) * 4);
    L.hK = w;     w += up16(hrows * L.hn * 4);
    L.hU = w;     w += up16(hrows * L.hn * 4);
    if (w - L.hK < up16((T * T + rows) * 4)) w = L.hK + up16((T * T + rows) * 4); // (the contraction's tables reuse the two blocks)
    L.rowacc = w; w += up16(rows * 4);
    L.dump = w;   w += ndump * (96 * 4);
    L.per_pair = w;
    L.total = o + (serial_slots ? serial_slots : BPP) * w;
    return L;
}

// One group (BHS steps) of a band's forward sweep, unrolled: the boundary row's entries of the group arrive in `hv` (one
// uniform 16-byte read per four steps) and reach lane 0 as the `old` operand of the DPP shift -- a lane without a source
// keeps it -- so the shift is one instruction; the store of the forward solution and the hand-over write take the step
// number as an immediate offset.
struct BandFwd {
    float cur, upprev, V, clo, kmax;
    int q1;
};
constexpr int BHS = 8; // steps per unrolled group (two per phase: the registers of 16 unrolled steps cost a wavefront per SIMD)
// FREEZE: a lane keeps its state once its row has ended -- the last band from the phase of its first row's end on, so that
// K[P][P] stays in its lane; everywhere else a lane past its row's end runs on (gamma = 0), which nobody reads.
template <bool FREEZE, bool COMP, bool GRAD>
__device__ __forceinline__ void bandp_fwd_phase(BandFwd &st, const float (&hv)[BHS], const float *dcrow, int n, unsigned qlim,
                                                float *ho, const float *wrow, int lane4)
{
    float gq[BHS]; // the group's increments up front (one LDS round trip per group instead of one per step on the dependent chain)
#pragma unroll
    for (int k = 0; k < BHS; ++k) gq[k] = dcrow[(st.q1 - 1 + k) >> n]; // (a column outside the grid: one of the row's zeros)
#pragma unroll
    for (int u = 0; u < BHS; ++u) {
        const bool active = !FREEZE || (unsigned)(st.q1 - 1) < qlim;
        const float g = gq[u];
        float up = hv[u]; // (lane 0 has no source lane: it keeps the boundary row's entry.  asm: the builtin copies hv[u] first)
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(up) : "v"(st.cur));
        // K11 - K01 = (K10 - K00) + F,  F = gamma (sqrt(3) t + gamma (t + K00)),  t = K10 + K01
        const float t = st.cur + up;
        float y = 1.7320508075688772f * t;
        y = __builtin_fmaf(t + st.upprev, g, y);
        const float Vn = __builtin_fmaf(g, y, st.V);
        float Vt = Vn, nlo = 0.f;
        if constexpr (COMP)
            Vt += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(st.clo), 0x138, 0xF, 0xF, true));
        const float nw = up + Vt;
        if constexpr (COMP) nlo = Vt - (nw - up);
        if (GRAD) asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(lane4), "v"(st.upprev), "s"(wrow), "n"(u * 256));
        ho[u] = COMP ? nw + nlo : nw;
        st.cur = active ? nw : st.cur;
        asm("v_max_f32 %0, |%1|, %0" : "+v"(st.kmax) : "v"(st.cur));
        if constexpr (COMP) st.clo = active ? nlo : st.clo;
        st.V = active ? Vn : st.V;
        st.upprev = active ? up : st.upprev;
        ++st.q1;
    }
}
// The same for the reverse sweep: the boundary lane is lane 63 of a full band (no DPP source from above: keeps hv[u]); the
// last band's boundary is U = 1, which lane L - 1 finds in lane L (a row outside the grid never leaves its initial 1).
struct BandRev {
    float cur, dprev, V, run, out, clo;
    int q;
};
// ALLIN: every lane is inside the grid on every step of the group (the top lanes' flush needs no column test).  The state of
// a lane outside the grid needs no protection in this direction (zeros around the rows, see band_zpad; nothing is read from it
// later), and neither do its block sums (see `run` below).
//
// Block sums of S = K_fwd * U: a lane sums its row over the r fine columns of a coarse cell (`run`), and the r lanes of a coarse
// row pass the cell's sum UP the lanes -- lane l + 1 ends a cell one step before lane l, so `out` = run + what the lane below
// handed over the step before (one DPP move and one multiply-add; `chain` = 0 in the bottom lane of a coarse row) is the
// cell's sum over the rows from l down on the step lane l ends it.  The TOP lanes of all coarse rows end their cells on the
// same steps, sp = L - 2 mod r -- and L - 2 = 6 mod 8 in every band (P and the band offsets are multiples of 8 <= r), i.e.
// always on step BFU = 6 of an unrolled group of 8: the group hands that step's sum and cell out, and the caller adds it to
// the fp64 table BEHIND the group on the groups that end a cell (r = 4, the coarsest refinement this kernel takes: L = 0 mod 4
// and the steps are 2 and 6 of every group; a uniform test per group; a branch inside the unrolled steps
// was measured: it splits the scheduling region and costs 8 %).  Rounds 3-4 / the serial kernel: one ds_add_f64 of every
// lane on every step, 8 of the step's 27 instructions.  The sum over a coarse row's lanes is fp32 (r <= 64 terms), the table
// stays fp64.
constexpr int BFU = 6;
constexpr int BKR = 8; // depth of the register ring the forward solution comes back through (16: maze shape -1.6 %, but 142 registers)
// COMPR: the reverse sweep's full-magnitude add in two floats as well (see COMP; one-channel launches: very smooth paths in one
// channel keep 1.8e-5 on the gradient with the forward add alone).
template <bool ALLIN, int H, bool COMPR>
__device__ __forceinline__ void bandp_rev_phase(BandRev &st, const float (&hv)[BHS], float (&kfr)[BKR], const float *rnext, int lanep,
                                                const float *dcrow, float chain, bool top, int n, int r, int P, bool rowvalid,
                                                float *ho, float &out6, int &cell6, float &out2, int &cell2)
{
    float gq[BHS];
#pragma unroll
    for (int k = 0; k < BHS; ++k) gq[k] = dcrow[(st.q - k) >> n]; // (a column outside the grid: one of the row's zeros)
#pragma unroll
    for (int u = 0; u < BHS; ++u) {
        const float g = gq[u];
        const float kf = kfr[(H + u) & (BKR - 1)];
        kfr[(H + u) & (BKR - 1)] = rnext[lanep - 64 * u];
        float down = hv[u]; // (lane 63 has no source lane: it keeps the boundary row's entry)
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(down) : "v"(st.cur));
        // (no activity test on K_fwd: a lane outside the grid sums garbage that nothing consumes -- ahead of its row the run is
        //  reset on column P = 0 mod r, the step before the row starts; behind it the run has been handed up on column 0; the
        //  bottom lane of a coarse row takes nothing from below, so rows outside the grid do not leak in)
        // S - 1 is what is summed: on smooth paths S = K_fwd U is 1 + O(increments) in every cell, the 4-corner scatter of the
        // block sums cancels the constant, and fp32 sums of r values near 1 would keep 6e-8 r of rounding each against a
        // difference of O(increments) (very smooth paths in one channel: 2.6e-5 on the gradient).  The constant comes back
        // where the scatter does not cancel it: the four corners of the point grid (the contraction's table) and the
        // condition number.
        st.run += __builtin_fmaf(kf, st.dprev, -1.f);
        const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(st.out), 0x130, 0xF, 0xF, true)); // lane 63: 0
        st.out = __builtin_fmaf(recv, chain, st.run);
        if (u == BFU) { // the one step of a group on which the top lanes can end a cell (see BFU): flushed behind the group
            out6 = st.out;
            cell6 = (top && (ALLIN || (unsigned)st.q < (unsigned)P)) ? st.q >> n : -1;
        }
        if (u == BFU - 4) { // (r = 4, grids of 65 .. 128 cells: the second such step of a group)
            out2 = st.out;
            cell2 = (top && (ALLIN || (unsigned)st.q < (unsigned)P)) ? st.q >> n : -1;
        }
        st.run = (st.q & (r - 1)) == 0 ? 0.f : st.run; // (a lane outside the grid carries run = 0 anyway)
        const float t = st.cur + down;
        float y = 1.7320508075688772f * t;
        y = __builtin_fmaf(t + st.dprev, g, y);
        const float Vn = __builtin_fmaf(g, y, st.V);
        float Vt = Vn, nlo = 0.f;
        if constexpr (COMPR) Vt += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(st.clo), 0x130, 0xF, 0xF, true));
        const float nw = down + Vt;
        if constexpr (COMPR) nlo = Vt - (nw - down);
        ho[BHS - 1 - u] = COMPR ? nw + nlo : nw; // (`ho` is the group's LAST entry: LDS offsets are unsigned.  A lane outside the grid writes into the padding, or entries nobody reads)
        if constexpr (COMPR) st.clo = nlo;
        st.cur = nw;
        st.V = Vn;
        st.dprev = down;
        --st.q;
    }
}

// SER: the launches with many pairs -- every wavefront of the workgroup owns a pair of its own (rows i .. i + nslots - 1 against
// the staged column) and walks its bands one after the other: the same groups of steps, no lag, no barrier inside a sweep,
// one boundary row reused in place (a band's writes trail its reads by 55 entries forwards, lead them by 7 or more backwards).
template <int DPAD, bool GRAD, bool SYM, bool COMP, bool SER>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(GRAD && SER ? 4 : 1, 4))) void gram_bandp_kernel(BandArgs a)
{
    extern __shared__ __align__(16) unsigned char band_smem[];
    const int tid = threadIdx.x, lane = tid & 63, NT = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, d = a.d, n = a.n, io64 = a.io64, Tm = T - 1, r = 1 << n;
    const int P = Tm << n, nb = (P + 63) >> 6, nsteps = P + 63;
    const int nslots = SER ? (int)(blockDim.x >> 6) : BPP;                              // pairs per workgroup
    const int slot = SER ? wave : __builtin_amdgcn_readfirstlane(wave / nb);           // pair of the workgroup
    const int band0 = SER ? 0 : wave - slot * nb;                                      // band of the pair (SER: all of them in turn)
    const bool lead = SER || band0 == 0;                                               // the wavefront with the pair's scalar work
    const double inv_h = a.inv_h;
    const float m2h = (float)(-2.0 * inv_h);
    const double dscale = 1.0 / ((double)r * (double)r * 3.46410161513775459); // 1 / (r^2 sqrt(12))
    const double inv_r2 = 1.0 / ((double)r * (double)r);
    const BandPLds lay = bandp_lds(T, P, DPAD, SER ? nslots : 0);
    const int BZP = band_zpad(r), DS = Tm + 2 * BZP; // zeros on either side of a row of the increment table, floats per row
    double *yd = reinterpret_cast<double *>(band_smem + lay.yd);
    double *yref = reinterpret_cast<double *>(band_smem + lay.yref);
    float *yf = reinterpret_cast<float *>(band_smem + lay.yf);
    unsigned char *pbase = band_smem + lay.pair0 + slot * lay.per_pair;
    double *Sc = reinterpret_cast<double *>(pbase + lay.Sc);          // [Tm][Tm] block sums of S = K_fwd * U
    float *Dc = reinterpret_cast<float *>(pbase + lay.Dc);            // [Tm + 1][DS] coarse increments between zeros (band_zpad); later the parked column sums [T][DPAD]
    float *hKall = reinterpret_cast<float *>(pbase + lay.hK);         // row b: K[64 (b + 1)][.], written by band b, read by band b + 1
    float *hUall = reinterpret_cast<float *>(pbase + lay.hU);         // row b: U[64 (b + 1)][.], written by band b + 1, read by band b
    float *rowacc = reinterpret_cast<float *>(pbase + lay.rowacc);
    float *misc = reinterpret_cast<float *>(pbase + lay.misc);        // [0..1]: K[P][P] (double), [4 + b]: grid maximum of band b
    float *ones = misc + 8;                                           // [8]: the boundary of the first / last band
    float *dump = reinterpret_cast<float *>(pbase + lay.dump + band0 * (96 * 4)); // [96]: a group writes 8 entries from the lane's cell
    float *wsw = GRAD ? a.wsk + ((size_t)blockIdx.x * nslots + slot) * a.wsk_per_wave + 32 * 64 : nullptr; // the pair's forward solution (32 rows of padding in front)
    const int ngf = (nsteps + BGS - 1) / BGS;               // forward phases of one band
    const int nsr = ngf * BGS;                              // rows of a band's forward-solution scratch (whole phases)
    const int ngr = (P + 63 + BGS - 1) / BGS;               // reverse phases of a full band (P + L - 1 <= P + 63 steps)
    const int Mf = ngf + BLAG * (nb - 1), Mr = ngr + BLAG * (nb - 1);

#ifdef SIGSVGD_PHASE_STAMPS
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#endif
    if (lead && lane < 8) ones[lane] = 1.f; // (read after the first pair's staging barriers)
    const long long it0 = a.nitems * blockIdx.x / gridDim.x, it1 = a.nitems * (blockIdx.x + 1) / gridDim.x;
    int remaining = (int)(it1 - it0);
    long long item = it0;
    int kq = 0, cstart = 0;
    {
        long long rem = it0;
        for (;; ++kq) {
            const int cn = a.B - (SYM ? a.tm.tile_of(kq) * nslots : 0);
            if (rem < cn) break;
            rem -= cn;
        }
        cstart = (int)rem;
    }
#pragma unroll 1
    while (remaining > 0) {
    const int itile = a.tm.tile_of(kq);
    const int cfirst = SYM ? itile * nslots : 0;
    const int ncolr = min(a.B - cfirst - cstart, remaining);
    const int i0 = itile * nslots, i = i0 + slot;
    const int j0 = cfirst + cstart, j1 = j0 + ncolr;
    const bool row_ok = i < a.A;
    if (GRAD && lead)
        for (int e = lane; e < T * DPAD; e += 64) rowacc[e] = 0.f;

#pragma unroll 1
    for (int j = j0; j < j1; ++j, ++item) {
        int lanep = lane;
        asm volatile("" : "+v"(lanep));
        // ---- stage y_j (coarse points, centred on its first point) ------------------------------------------------
        __syncthreads();
        for (int e = tid; e < T * DPAD; e += NT) {
            const int t = e / DPAD, c = e % DPAD;
            const double r0 = c < d ? b_ldany(a.Y, (size_t)j * T * d + c, io64) : 0.0;
            const double v = c < d ? b_ldany(a.Y, ((size_t)j * T + t) * d + c, io64) - r0 : 0.0;
            yd[t * (DPAD + 1) + c] = v;
            yf[t * DPAD + c] = (float)v;
            if (t == 0) yref[c] = r0;
            double s = v * v;
#pragma unroll
            for (int off = 1; off < DPAD; off <<= 1) s += __shfl_xor(s, off, 64);
            if (c == 0) yd[t * (DPAD + 1) + DPAD] = -s * inv_h;
        }
        __syncthreads();

        SIGB_STAMP(0)
        const bool valid = row_ok && (!SYM || j >= i); // (uniform per wavefront)
        if (valid && lead) {
            for (int e = lanep; e < (Tm + 1) * DS; e += 64) Dc[e] = 0.f; // the zeros around the rows (the table is reused per pair)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // ---- coarse static kernel: lane m = point row m; G[m][b] in fp64, row differences, 4-corner increments --------
            const int m = min(lanep, T - 1);
            double xs[DPAD], xn = 0.0;
#pragma unroll
            for (int c = 0; c < DPAD; ++c) {
                const double xc = c < d ? b_ldany(a.X, ((size_t)i * T + m) * d + c, io64) - yref[c] : 0.0;
                xn = __builtin_fma(xc, xc, xn);
                xs[c] = xc * (2.0 * inv_h);
            }
            xn = -xn * inv_h;
            double gprev = 0.0;
            for (int b = 0; b < T; ++b) {
                const double *yr = yd + b * (DPAD + 1);
                double e2 = xn + yr[DPAD];
#pragma unroll
                for (int c = 0; c < DPAD; ++c) e2 = __builtin_fma(xs[c], yr[c], e2);
                const double g = exp64(e2);
                const double rd = g - gprev; // G[m][b] - G[m][b-1]
                gprev = g;
                const double nbr = shfl_down_f64(rd); // row m + 1
                if (b >= 1 && lanep < Tm) Dc[lanep * DS + BZP + (b - 1)] = (float)((nbr - rd) * dscale);
            }
            if (GRAD)
                for (int e = lanep; e < Tm * Tm; e += 64) Sc[e] = 0.0;
        }
        SIGB_STAMP(1)
        __syncthreads(); // the pair's increment table is complete
        SIGB_STAMP(3)

        // ---- forward sweep: this wavefront's band, BGS steps per phase ---------------------------------------------
        float kmax = 1.f; // largest |K| this lane has seen on the pair's grid
        int band = band0;
        do { // (SER: the bands in turn; otherwise this wavefront's band, once -- a compile-time fact, not a loop)
            const int p = 64 * band + lanep;
            const bool rowvalid = p < P;
            const float *dcrow = Dc + (rowvalid ? (p >> n) : Tm) * DS + BZP; // (a row outside the grid: the row of zeros)
            const float *wb = GRAD ? wsw + (size_t)band * nsr * 64 : nullptr;
            BandFwd st;
            st.cur = 1.f; st.upprev = 1.f; st.V = 0.f; st.clo = 0.f; st.kmax = kmax;
            st.q1 = 1 - lanep; // column + 1 of the cell in work
            // lane 63 hands K[64 band + 64][q + 1] over through entry q + 1 of row `band` (entry e at [BPAD - 1 + e]: the 16
            // entries a phase of the next band reads start on a 16-byte boundary); the other lanes write into the dump block
            const bool hands = lanep == 63 && band < nb - 1;
            float *ho = hands ? hKall + (SER ? 0 : band) * lay.hn + (BPAD - 1) + st.q1 : dump + lanep;
            const int hinc = hands ? BHS : 0; // (per group of steps)
            const float *hin = hKall + (SER ? 0 : band - 1) * lay.hn + BPAD; // (band 0: not read)
            const unsigned qlim = rowvalid ? (unsigned)P : 0u;
#pragma unroll 1
            for (int ph = 0; ph < (SER ? ngf : Mf); ++ph) {
                const int gi = SER ? ph : ph - BLAG * band;
                if (valid && gi >= 0 && gi < ngf) {
                    const int s0 = gi * BGS;
                    // lane 0's upper neighbour on step s is entry s + 1 of the row band - 1 leaves: written on ITS step s + 63,
                    // i.e. the 16 entries of this phase by step s0 + 78 < (gi + BLAG) BGS, the steps the neighbour has finished
                    const bool freeze = band == nb - 1 && s0 + BGS > P; // (the last band, once its first row may have ended)
#pragma unroll 1
                    for (int h = 0; h < BGS; h += BHS) {
                        float hv[BHS];
                        { // (band 0: eight ones kept in LDS -- a select of the address instead of a branch around the loads)
                            const float4 *h4 = reinterpret_cast<const float4 *>(band ? hin + s0 + h : ones);
#pragma unroll
                            for (int k = 0; k < BHS / 4; ++k) {
                                const float4 v = h4[k];
                                hv[4 * k] = v.x; hv[4 * k + 1] = v.y; hv[4 * k + 2] = v.z; hv[4 * k + 3] = v.w;
                            }
                        }
                        if (band && s0 + h + BHS > P) { // entries beyond P were never written: their lanes are past the grid, but
#pragma unroll                                            // whatever they read reaches the grid maximum (kmax)
                            for (int k = 0; k < BHS; ++k)
                                if (s0 + h + k + 1 > P) hv[k] = 1.f;
                        }
                        const float *wrow = GRAD ? wb + (size_t)(s0 + h) * 64 : nullptr;
                        if (freeze)
                            bandp_fwd_phase<true, COMP, GRAD>(st, hv, dcrow, n, qlim, ho, wrow, lanep * 4);
                        else
                            bandp_fwd_phase<false, COMP, GRAD>(st, hv, dcrow, n, qlim, ho, wrow, lanep * 4);
                        ho += hinc;
                    }
                    SIGB_STAMP(2)
                }
                if (!SER) __syncthreads();
                SIGB_STAMP(3)
            }
            kmax = st.kmax;
            if (valid && band == nb - 1 && p == P - 1) *reinterpret_cast<double *>(misc) = (double)st.cur + (double)st.clo;
        } while (SER && ++band < nb);
        if (valid) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) kmax = fmaxf(kmax, __shfl_xor(kmax, off, 64));
            if (lanep == 0) misc[4 + band0] = kmax;
        }
        __syncthreads();
        double kfin = 1.0;
        float kfin_keep = 0.f;
        bool canc_keep = false;
        if (valid && lead) {
            kfin = *reinterpret_cast<const double *>(misc);
            float km = misc[4];
            for (int b = 1; b < (SER ? 1 : nb); ++b) km = fmaxf(km, misc[4 + b]);
            const float kfv = (float)kfin;
            const bool cancelled = kfv == kfv && km > (d == 1 ? 2.f : (d == 2 || !COMP) ? 4.f : 8.f) * fmaxf(fabsf(kfv), 0.1f);
            if (lanep == 0) {
                b_stany(a.K, (size_t)i * a.B + j, kfin, io64);
                if (SYM && j != i) b_stany(a.K, (size_t)j * a.B + i, kfin, io64);
                if (!GRAD) a.kflag[(size_t)i * a.B + j] = cancelled ? 1 : 0;
            }
            kfin_keep = kfv;
            canc_keep = cancelled;
        }

        if (GRAD) {
            // ---- reverse sweep: the last band leads ---------------------------------------------------------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this band's forward solution has left the wavefront
            band = SER ? nb - 1 : band0;
            do {
                const int p = 64 * band + lanep;
                const bool rowvalid = p < P;
                const int L = min(64, P - 64 * band);
                const int arow = min(p, P - 1) >> n;
                const float *dcrow = Dc + (rowvalid ? arow : Tm) * DS + BZP;
                double *scrow = Sc + arow * Tm;
                const bool lastband = band == nb - 1;
                BandRev st;
                st.cur = 1.f; st.dprev = 1.f; st.V = 0.f; st.run = 0.f; st.out = 0.f; st.clo = 0.f;
                st.q = P - 1 + (L - 1 - lanep);
                const int rr = min(r, 64);
                const float chain = ((p & (rr - 1)) == rr - 1) ? 0.f : 1.f; // the bottom lane of a coarse row takes nothing from below
                const bool top = rowvalid && (p & (rr - 1)) == 0;
                // K_fwd[p][q] was stored on forward step lane + q: row R = P + L - 2 - sp of the band's scratch on step sp
                const float *wrow = wsw + (size_t)band * nsr * 64 + lanep;
                const int R = P + L - 2;
                const float *rnext = wsw + ((size_t)band * nsr + (R - BKR)) * 64; // (uniform row pointer of the ring's next load)
                // lane 0 hands U[64 band][q] over through entry q of row band - 1 (entry e at [BUO + e])
                const bool hands = lanep == 0 && band > 0;
                float *ho = hands ? hUall + (SER ? 0 : band - 1) * lay.hn + BUO + st.q : dump + 16 + lanep;
                const int hinc = hands ? -BHS : 0; // (per group of steps)
                const float *hin = hUall + (SER ? 0 : band) * lay.hn + BUO + (P - BGS); // (last band: not read)
                float kfr[BKR];
                if (valid) {
#pragma unroll
                    for (int u = 0; u < BKR; ++u) kfr[u] = wrow[(size_t)max(R - u, 0) * 64];
                } else {
#pragma unroll
                    for (int u = 0; u < BKR; ++u) kfr[u] = 0.f;
                }
                const int ngb = (P + L - 1 + BGS - 1) / BGS; // this band's phases
                const int rb = nb - 1 - band;
#pragma unroll 1
                for (int ph = 0; ph < (SER ? ngb : Mr); ++ph) {
                    const int gi = SER ? ph : ph - BLAG * rb;
                    if (valid && gi >= 0 && gi < ngb) {
                        const int sp0 = gi * BGS;
                        // lane 63's lower neighbour on step sp is entry P - 1 - sp of the row band + 1 leaves (its lane 0 on
                        // its step L' - 1 + sp): the 16 entries of this phase are there once it has finished step sp0 + L' + 14
                        const bool plat = L == 64 && sp0 >= 64 && sp0 + BGS <= P; // every lane inside the grid on every step
#pragma unroll 1
                        for (int h = 0; h < BGS; h += BHS) {
                            float hv[BHS];
                            {
                                // entries P - 8 - (sp0 + h) .. P - 1 - (sp0 + h), descending over the group's steps (last band: ones)
                                const float4 *h4 = reinterpret_cast<const float4 *>(lastband ? ones : hin + (BGS - BHS) - sp0 - h);
#pragma unroll
                                for (int k = 0; k < BHS / 4; ++k) {
                                    const float4 v = h4[k];
                                    hv[BHS - 1 - 4 * k] = v.x; hv[BHS - 2 - 4 * k] = v.y; hv[BHS - 3 - 4 * k] = v.z; hv[BHS - 4 - 4 * k] = v.w;
                                }
                            }
                            if (!lastband && sp0 + h + BHS > P) { // entries below 0 were never written (lanes past the grid)
#pragma unroll
                                for (int k = 0; k < BHS; ++k)
                                    if (sp0 + h + k > P - 1) hv[k] = 1.f;
                            }
                            float out6, out2;
                            int cell6, cell2;
                            static_assert(BKR == BHS, "a deeper ring needs the group's position in it as a compile-time constant");
                            if (COMP && a.comprev) // (one channel, or order 4 on 160 cells and more: band_comp)
                                bandp_rev_phase<false, 0, true>(st, hv, kfr, rnext, lanep, dcrow, chain, top, n, r, P, rowvalid, ho - (BHS - 1), out6, cell6, out2, cell2);
                            else if (plat)
                                bandp_rev_phase<true, 0, false>(st, hv, kfr, rnext, lanep, dcrow, chain, top, n, r, P, rowvalid, ho - (BHS - 1), out6, cell6, out2, cell2);
                            else
                                bandp_rev_phase<false, 0, false>(st, hv, kfr, rnext, lanep, dcrow, chain, top, n, r, P, rowvalid, ho - (BHS - 1), out6, cell6, out2, cell2);
                            rnext -= BHS * 64;
                            ho += hinc;
                            // (step BFU of the group ends the top lanes' cells iff L - 2 - (sp0 + h + BFU) = 0 mod r)
                            if (r == 4 && cell2 >= 0) unsafeAtomicAdd(scrow + cell2, (double)out2); // (L = 0 mod 4: steps 2 and 6 of every group)
                            if (((L - 2 - sp0 - h - BFU) & (r - 1)) == 0 && cell6 >= 0) unsafeAtomicAdd(scrow + cell6, (double)out6); // ds_add_f64
                        }
                        SIGB_STAMP(4)
                    }
                    if (!SER) __syncthreads();
                    SIGB_STAMP(5)
                }
                // (the ring's last loads are never used: consumed here, or hipcc carries them as pending into the next pair's
                //  forward step loop and waits for vmcnt(0) in every step -- behind the step's own store, a memory round trip)
#pragma unroll
                for (int u = 0; u < BKR; ++u) asm volatile("" ::"v"(kfr[u]));
            } while (SER && --band >= 0);

            // ---- after the sweeps: the coarse contraction, spread over the pair's wavefronts -------------------------------
            // R = 4-corner scatter of S_coarse / r^2 as an fp32 table and x~ in fp32, staged by all of them in the boundary
            // rows' LDS (free now); then the row side on band 0's wavefront and the column side on band 1's, side by side.
            float *Rt = hKall;         // [T][T]
            float *xl = hKall + T * T; // [T][DPAD]
            if (valid) {
                auto Sat = [&](int aa, int bb) -> float {
                    return (aa >= 0 && aa < Tm && bb >= 0 && bb < Tm) ? (float)(Sc[aa * Tm + bb] * inv_r2) : 0.f;
                };
                const int wtid = band0 * 64 + lanep, wnt = SER ? 64 : nb * 64;
                for (int e = wtid; e < T * T; e += wnt) {
                    const int m = e / T, nn = e - m * T;
                    // (the table holds sums of S - 1: + r^2 per cell inside the grid, i.e. +-1 at the four corners after the scatter)
                    const float corner = ((m == 0 || m == Tm) && (nn == 0 || nn == Tm)) ? (m == nn ? 1.f : -1.f) : 0.f;
                    Rt[e] = (Sat(m - 1, nn - 1) + Sat(m, nn)) - (Sat(m - 1, nn) + Sat(m, nn - 1)) + corner;
                }
                for (int e = wtid; e < T * DPAD; e += wnt) {
                    const int m = e / DPAD, c = e - m * DPAD;
                    xl[e] = c < d ? (float)(b_ldany(a.X, ((size_t)i * T + m) * d + c, io64) - yref[c]) : 0.f;
                }
            }
            if (valid && lead) {
                // the pair's verdict for the exact fp64 pass
                bool ill = false;
                if (d <= 3) {
                    float cs = 0.f;
                    for (int e = lanep; e < Tm * Tm; e += 64) {
                        const int ra = e / Tm;
                        cs = __builtin_fmaf(fabsf((float)(Sc[e] + (double)(r * r))), fabsf(Dc[ra * DS + BZP + (e - ra * Tm)]), cs);
                    }
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) cs += __shfl_xor(cs, off, 64);
                    ill = kfin_keep == kfin_keep && cs * 3.46410161513775459f > 150.f * fmaxf(fabsf(kfin_keep), 0.1f);
                }
                if (lanep == 0) a.kflag[(size_t)i * a.B + j] = (canc_keep || ill) ? 1 : 0;
            }
            __syncthreads(); // the tables are staged; the increments (verdict) are not needed any more
            const int colwave = nb > 1 ? 1 : 0; // the wavefront of the column side (a one-band pair has one wavefront for both)
            if (valid && (SER || band0 == 0 || (SYM && band0 == colwave))) {
                float w_ij = 1.f, w_ji = 1.f;
                if (a.go) {
                    w_ij = (float)b_ldany(a.go, (size_t)i * a.B + j, io64);
                    if (SYM || a.symw) w_ji = (float)b_ldany(a.go, (size_t)j * a.B + i, io64);
                    if (a.symw) { w_ij += w_ji; w_ji = w_ij; }
                } else if (a.symw) {
                    w_ij = 2.f; w_ji = 2.f;
                }
                if (SYM && j == i) w_ji = 0.f; // diagonal pair: first-slot derivative only
                const float ns32 = (float)(-inv_h * 1.4426950408889634074);
                const int me = min(lanep, T - 1);
                float acc[DPAD];
#pragma unroll
                for (int c = 0; c < DPAD; ++c) acc[c] = 0.f;
                if (lead) { // row side: lane m sums over the columns n
                    float xf[DPAD];
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) xf[c] = xl[me * DPAD + c];
                    for (int nn = 0; nn < T; ++nn) {
                        const float Rv = Rt[me * T + nn];
                        const float *yr = yf + nn * DPAD;
                        float df[DPAD], e2 = 0.f;
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) {
                            df[c] = xf[c] - yr[c];
                            e2 = __builtin_fmaf(df[c], df[c], e2);
                        }
                        const float rg = Rv * __builtin_amdgcn_exp2f(e2 * ns32);
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) acc[c] = __builtin_fmaf(rg, df[c], acc[c]);
                    }
                    if (lanep < T) {
#pragma unroll
                        for (int c = 0; c < DPAD; ++c)
                            if (c < d) rowacc[me * DPAD + c] += w_ij * m2h * acc[c];
                    }
                }
                if (SYM && (SER || band0 == colwave)) { // column side (Y is X): lane n sums over the rows m; the sums are parked in the increment table's place
#pragma unroll
                    for (int c = 0; c < DPAD; ++c) acc[c] = 0.f;
                    const float *yr = yf + me * DPAD;
                    for (int m = 0; m < T; ++m) {
                        const float Rv = Rt[m * T + me];
                        const float *xr = xl + m * DPAD;
                        float df[DPAD], e2 = 0.f;
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) {
                            df[c] = xr[c] - yr[c];
                            e2 = __builtin_fmaf(df[c], df[c], e2);
                        }
                        const float rg = Rv * __builtin_amdgcn_exp2f(e2 * ns32);
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) acc[c] = __builtin_fmaf(rg, df[c], acc[c]);
                    }
                    if (lanep < T) { // d k(x_j, x_i) / d y_n = -(2/h) sum_m R G (y~_n - x~_m)
#pragma unroll
                        for (int c = 0; c < DPAD; ++c) Dc[lanep * DPAD + c] = -(w_ji * m2h) * acc[c];
                    }
                }
            } else if (SYM && !valid && lead) {
                for (int e = lane; e < T * DPAD; e += 64) Dc[e] = 0.f; // no pair in this slot: nothing to add to the column
            }
        }

        SIGB_STAMP(6)
        if (GRAD && SYM) {
            __syncthreads(); // every pair has parked its column-side sums
            float *dstc = a.cslab + (size_t)item * (T * d);
            for (int e = tid; e < T * d; e += NT) {
                const int nn = e / d, c = e - nn * d;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < nslots; ++w)
                    s += reinterpret_cast<const float *>(band_smem + lay.pair0 + w * lay.per_pair + lay.Dc)[nn * DPAD + c];
                dstc[e] = s;
            }
        }
    }
    if (GRAD && row_ok && lead) { // the segment's row-side sums
        const int tot = T * d;
        double *dstr = a.rseg + (((size_t)(kq + (int)blockIdx.x)) * nslots + slot) * (size_t)tot;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int e = lane; e < tot; e += 64) {
            const int m = e / d, c = e - m * d;
            dstr[e] = (double)rowacc[m * DPAD + c];
        }
    }
    remaining -= ncolr;
    ++kq;
    cstart = 0;
    } // row tiles of the range
#ifdef SIGSVGD_PHASE_STAMPS
    SIGB_STAMP(0)
    if (lane == 0 && a.stamps)
        for (int k = 0; k < 8; ++k) atomicAdd(&a.stamps[k], ph_[k]);
#endif
}

} // namespace

bool band_supported(int A, int B, int T, int d, int n, int kind, unsigned flags)
{
    (void)A; (void)B;
    if (n < 1 || n > 7 || T < 3 || T > BTMAX || d > 16) return false;
    const int P = (T - 1) << n;
    if (P <= 128 || P > BPMAX) return false; // (so r = P / (T - 1) >= 8: the kernels' unrolled groups rely on it)
    if (kind != SIGSVGD_STATIC_RBF) return false;
    if (flags & SIGSVGD_FLAG_NAIVE_SOLVER) return false;
    return true;
}

namespace {
inline size_t band_flag_bytes(int A, int B) { return (((size_t)A * B + 255) & ~(size_t)255) + generic_repair_bytes(); }

// pairs per workgroup of the serial schedule: eight (two wavefronts per SIMD) where LDS holds them
inline int band_serial_slots(int T, int d, int n)
{
    const int P = (T - 1) << n, dpad = d <= 8 ? 8 : 16;
    const BandPLds one = bandp_lds(T, P, dpad, 1);
    const int fit = (158 * 1024 - one.pair0) / one.per_pair;
    return fit >= 8 ? 8 : (fit < 1 ? 1 : fit);
}
// workgroups a CU holds: by LDS (160 KB a CU) and by wavefronts (every instantiation fits four per SIMD: <= 128 registers)
inline int band_wg_per_cu(int T, int d, int n, bool serial)
{
    const int P = (T - 1) << n, nb = (P + 63) >> 6, dpad = d <= 8 ? 8 : 16;
    const int slots = serial ? band_serial_slots(T, d, n) : 0;
    const BandPLds L = bandp_lds(T, P, dpad, slots);
    const int by_lds = (160 * 1024) / (L.total + 1024);
    const int by_waves = 16 / (serial ? slots : BPP * nb);
    const int k = by_lds < by_waves ? by_lds : by_waves;
    return k < 1 ? 1 : k;
}
inline GradGeom band_geometry(int A, int B, int T, int d, int n, bool sym, bool serial)
{
    return grad_geometry(A, B, T * d, sym, 0, 1, false, serial ? band_serial_slots(T, d, n) : BPP,
                         (long long)device_cu_count() * band_wg_per_cu(T, d, n, serial));
}
inline size_t band_wsk_per_pair(int T, int n)
{
    const int P = (T - 1) << n;
    const size_t rows = (size_t)((P + 63 + BGS - 1) / BGS) * BGS; // whole phases per band
    return (size_t)((P + 63) >> 6) * rows * 64 + 32 * 64; // floats (+ 32 rows in front: the ring's loads need no clamp)
}
// Which schedule a launch takes.  Band-parallel wins while its workgroups (one pair each) pass through the chip in a few
// rounds -- its wavefronts idle BLAG (nb - 1) phases of every sweep, which other workgroups on the CU fill, but the sum of a
// pair's wavefront time is nb / (1 + BLAG (nb - 1) BGS / (P + 63)) times the serial schedule's.  Measured (Gram + gradient,
// symmetric, ms, parallel / serial): 10 points order 4 (3 bands, 1,280 resident workgroups) -- N = 50 / 70 / 100 / 150: 0.105 /
// 0.189 / 0.364 / 0.78 against 0.125 / 0.205 / 0.424 / 0.737; 30 points order 3 (4 bands, 1,024) -- N = 35 / 60 / 100: 0.138 /
// 0.317 / 0.832 against 0.239 / 0.300 / 0.835.  Rule: at most five rounds with three bands, one and a half with four.  With
// two bands (65 .. 128 cells) it beats the refined-grid kernel of gram_dyad.hip at every size measured (N = 64 .. 400: 20 points
// order 2 0.156 / 0.423 / 1.52 against 0.170 / 0.477 / 1.78; 5 points order 5 0.80 / 3.13 against 0.92 / 3.52): always.
// SIGSVGD_BAND_MODE=serial|parallel (read per launch) overrides it: the tests drive both schedules over the same shapes.
inline bool band_rule_parallel(int A, int B, int T, int d, int n, bool sym)
{
    const char *e = getenv("SIGSVGD_BAND_MODE");
    if (e && e[0] == 's') return false;
    if (e && e[0] == 'p') return true;
    const long long pairs = sym ? (long long)A * (A + 1) / 2 : (long long)A * B;
    const int nb = (((T - 1) << n) + 63) >> 6;
    if (nb <= 2) return true;
    return 2 * pairs <= (nb >= 4 ? 3ll : 10ll) * device_cu_count() * band_wg_per_cu(T, d, n, false);
}
inline bool band_use_parallel(const GramProblem &p, bool sym) { return band_rule_parallel(p.A, p.B, p.T, p.d, p.n, sym); }
// (the serial schedule takes grids of 129 .. 256 cells only: smaller ones that are not band-parallel stay on gram_dyad.hip)
inline bool band_is_serial(int A, int B, int T, int d, int n, bool sym)
{
    return ((T - 1) << n) > 128 && !band_rule_parallel(A, B, T, d, n, sym);
}
// forward-solution scratch: one block per pair a launch of `grid` workgroups has in flight
inline size_t band_wsk_bytes(int grid, int T, int d, int n, bool serial)
{
    const size_t pairs = (size_t)(grid > 0 ? grid : 1) * (serial ? band_serial_slots(T, d, n) : BPP);
    return ((pairs * band_wsk_per_pair(T, n) * sizeof(float)) + 255) & ~(size_t)255;
}
// bytes a launch of this shape and orientation needs, on the schedule the launcher will pick for it
inline size_t band_need_bytes(int A, int B, int T, int d, int n, int want_grad, bool sym)
{
    if (!want_grad) return band_flag_bytes(A, B) + 512;
    const bool serial = band_is_serial(A, B, T, d, n, sym);
    const GradGeom g = band_geometry(A, B, T, d, n, sym, serial);
    return g.rseg_bytes + g.cslab_bytes + band_wsk_bytes(g.grid, T, d, n, serial) + band_flag_bytes(A, B) + 1024;
}
} // namespace

int band_workspace_bytes(int A, int B, int T, int d, int n, int want_grad, unsigned flags, size_t *bytes)
{
    // a query with SIGSVGD_FLAG_Y_IS_X is the symmetric launch (half the pairs: often the band-parallel schedule, whose
    // scratch is a third of the serial one's); without the flag it covers both orientations
    const size_t yb = A == B ? band_need_bytes(A, B, T, d, n, want_grad, true) : 0;
    if ((flags & SIGSVGD_FLAG_Y_IS_X) && A == B) {
        *bytes = yb;
        return SIGSVGD_OK;
    }
    const size_t ob = band_need_bytes(A, B, T, d, n, want_grad, false);
    *bytes = ob > yb ? ob : yb;
    return SIGSVGD_OK;
}

namespace {
// (the dynamic-LDS limit of an instantiation, raised once per device: see gram_generic.hip)
template <int DPAD, bool GRAD, bool SYM, bool COMP, bool SER>
hipError_t band_raise_lds()
{
    static std::atomic<unsigned long long> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !((raised.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_bandp_kernel<DPAD, GRAD, SYM, COMP, SER>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) raised.fetch_or(1ull << dev, std::memory_order_release);
    }
    return hipSuccess;
}
// Where the full-magnitude add of a step runs in two floats (template parameter COMP; `rev`: in the reverse sweep too).  The
// add's rounding has the same sign row after row when neighbouring cells have nearly identical increments, so K drifts by
// up to 6e-8 per ROW on smooth paths: dyadic order >= 5 (forward sweep: K 1.2e-5 at order 6 without it); one channel at any
// order and order 4 from 160 cells on in both sweeps (very smooth paths, |step|^2 / h ~ 4e-5: the gradient -- 1e-3 of K there --
// was 3e-5 in one channel and 1.03e-5 at 208 cells, 6.5e-6 at 160, in two in the refined soak).  The notebook's (144 cells, order 4) and the maze script's (order
// 3) shapes stay without it: 5e-6 at worst in the sweep of profiles/r04_precision_sweep_refined.md.
inline bool band_comp(int T, int d, int n, bool *rev)
{
    const int P = (T - 1) << n;
    const bool both = d == 1 || (n == 4 && P >= 160);
    if (rev) *rev = both;
    return n >= 5 || both;
}
template <int DPAD, bool SER>
int band_launch_variant(const GramProblem &p, BandArgs &a, const GradGeom &g, bool grad, bool sym)
{
    if (g.tm.owned <= 0 || g.nitems <= 0) return SIGSVGD_OK;
    a.tm = g.tm;
    a.nitems = g.nitems;
    const int P = (p.T - 1) << p.n, nb = (P + 63) >> 6;
    const int slots = SER ? band_serial_slots(p.T, p.d, p.n) : 0;
    dim3 grid((unsigned)g.grid), block((SER ? slots : BPP * nb) * 64);
    const bool comp = band_comp(p.T, p.d, p.n, nullptr);
    const unsigned lds = (unsigned)bandp_lds(p.T, P, DPAD, slots).total;
#define SIGB_LAUNCH(G, S)                                                                                          \
    {                                                                                                              \
        hipError_t ae = comp ? band_raise_lds<DPAD, G, S, true, SER>() : band_raise_lds<DPAD, G, S, false, SER>(); \
        if (ae != hipSuccess) return hip_fail(ae, "hipFuncSetAttribute(gram_bandp_kernel)");                       \
        if (comp) hipLaunchKernelGGL((gram_bandp_kernel<DPAD, G, S, true, SER>), grid, block, lds, p.stream, a);   \
        else hipLaunchKernelGGL((gram_bandp_kernel<DPAD, G, S, false, SER>), grid, block, lds, p.stream, a);       \
    }
    if (grad && sym)
        SIGB_LAUNCH(true, true)
    else if (grad)
        SIGB_LAUNCH(true, false)
    else if (sym)
        SIGB_LAUNCH(false, true)
    else
        SIGB_LAUNCH(false, false)
#undef SIGB_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "launch gram_bandp_kernel");
    return SIGSVGD_OK;
}
} // namespace

// Refined grids of 65 .. 128 cells per side (two bands) with r >= 4 -- BASELINE C1, the planning script's shape: the
// band-parallel schedule takes them from the refined-grid kernel of gram_dyad.hip (which keeps grids of exactly 64 cells and
// dyadic order 1; SIGSVGD_BAND_MODE=serial sends them back to it: the tests compare the two).
bool band_takes_refined(const GramProblem &p)
{
    if (p.n < 2 || p.n > 7 || p.T < 3 || p.T > BTMAX || p.d > 16) return false;
    const int P = (p.T - 1) << p.n;
    if (P < 64 || P > 128 || (P == 64 && p.d != 1)) return false; // (64 cells, one band: only for the one-channel launches, whose
                                                                      //  very smooth regime wants the two-float add in both sweeps)
    if (p.kind != SIGSVGD_STATIC_RBF || (p.flags & (SIGSVGD_FLAG_NAIVE_SOLVER | SIGSVGD_FLAG_FORCE_GENERIC))) return false;
    return band_use_parallel(p, (p.flags & SIGSVGD_FLAG_Y_IS_X) && p.A == p.B);
}
size_t band_refined_workspace_bytes(int A, int B, int T, int d, int n, int want_grad, unsigned flags)
{
    size_t bytes = 0;
    (void)band_workspace_bytes(A, B, T, d, n, want_grad, flags, &bytes);
    return bytes;
}

int band_launch(const GramProblem &p)
{
    const bool grad = p.gradX_out != nullptr;
    const bool sym = (p.flags & SIGSVGD_FLAG_Y_IS_X) && p.A == p.B;
    // (grids of up to 128 cells come here for the band-parallel schedule only: band_takes_refined)
    const bool serial = band_is_serial(p.A, p.B, p.T, p.d, p.n, sym);
    BandArgs a;
    a.X = p.X; a.Y = p.Y; a.go = p.grad_out; a.K = p.K_out; a.rseg = nullptr; a.cslab = nullptr; a.wsk = nullptr;
    a.wsk_per_wave = band_wsk_per_pair(p.T, p.n);
    a.io64 = p.dtype == SIGSVGD_F64; a.A = p.A; a.B = p.B; a.T = p.T; a.d = p.d; a.n = p.n;
    a.symw = (p.flags & SIGSVGD_FLAG_SYM) ? 1 : 0; a.inv_h = p.inv_h;
    {
        bool rev = false;
        (void)band_comp(p.T, p.d, p.n, &rev);
        a.comprev = rev ? 1 : 0;
    }
    a.nitems = 0;
    if (a.symw && p.A != p.B) {
        set_error("sym backward needs A == B");
        return SIGSVGD_E_BADARG;
    }
    const GradGeom g = band_geometry(p.A, p.B, p.T, p.d, p.n, sym, serial);
    const size_t slabs = grad ? (g.rseg_bytes + g.cslab_bytes + 255) & ~(size_t)255 : 0;
    const size_t need = band_flag_bytes(p.A, p.B) + slabs + (grad ? band_wsk_bytes(g.grid, p.T, p.d, p.n, serial) : 0) + 256;
    if (!p.ws || p.ws_bytes < need) {
        set_error("band: workspace %zu B < required %zu B", p.ws_bytes, need);
        return SIGSVGD_E_WORKSPACE;
    }
    unsigned char *base = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(p.ws) + 255) & ~(uintptr_t)255);
    a.kflag = base;
    base += band_flag_bytes(p.A, p.B);
    if (grad) {
        a.rseg = reinterpret_cast<double *>(base);
        a.cslab = sym ? reinterpret_cast<float *>(base + g.rseg_bytes) : nullptr;
        a.wsk = reinterpret_cast<float *>(base + slabs);
    }
#ifdef SIGSVGD_PHASE_STAMPS
    {
        static unsigned long long *dbg = nullptr;
        if (!dbg) (void)hipMalloc(&dbg, 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbg, 0, 8 * sizeof(unsigned long long), p.stream);
        a.stamps = dbg;
    }
#endif
    int rc;
    if (serial)
        rc = p.d <= 8 ? band_launch_variant<8, true>(p, a, g, grad, sym) : band_launch_variant<16, true>(p, a, g, grad, sym);
    else
        rc = p.d <= 8 ? band_launch_variant<8, false>(p, a, g, grad, sym) : band_launch_variant<16, false>(p, a, g, grad, sym);
    if (rc) return rc;
#ifdef SIGSVGD_PHASE_STAMPS
    {
        unsigned long long hs[8];
        (void)hipStreamSynchronize(p.stream);
        (void)hipMemcpy(hs, a.stamps, sizeof(hs), hipMemcpyDeviceToHost);
        double tot = 0;
        for (int k = 0; k < 8; ++k) tot += (double)hs[k];
        static const char *nm[8] = {"staging/other", "static kernel", "forward steps", "forward barriers + idle phases",
                                    "reverse steps", "reverse barriers + idle phases", "verdict + coarse gradient / wait", "-"};
        fprintf(stderr, "[phase stamps band %s] A=%d T=%d d=%d n=%d grad=%d sym=%d: ", serial ? "serial" : "parallel", p.A, p.T, p.d,
                p.n, grad ? 1 : 0, sym ? 1 : 0);
        for (int k = 0; k < 7; ++k) fprintf(stderr, "%s %.1f%% | ", nm[k], 100.0 * (double)hs[k] / tot);
        fprintf(stderr, "total %.3e wave-cycles\n", tot);
    }
#endif
    // fp64 pass of the coverage kernel over the flagged pairs (a few microseconds when there are none)
    rc = generic_repair_launch(p, a.kflag, nullptr, sym, g.tm, g.NW);
    if (rc || !grad) return rc;
    return grad_reduce_launch(g, a.rseg, a.cslab, p.gradX_out, p.dtype == SIGSVGD_F64, p.A, p.B, p.T * p.d, sym, p.stream);
}

} // namespace sigsvgd
