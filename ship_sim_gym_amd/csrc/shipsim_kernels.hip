// shipsim_kernels.hip — hand-written gfx950 (CDNA4 / MI355X) kernels for the batched ShipEnv hot path.
//
// One wavefront lane per env.  Body state lives as FP64 struct-of-arrays columns in HBM (lane-contiguous,
// coalesced 8-byte loads/stores); the map bank (river-bank hull planes + goal centres) is staged in LDS once
// per workgroup; there is no dense contraction anywhere, so no MFMA.  The arithmetic follows the reference's
// operation order (pymunk 5.4.0 / Chipmunk2D cpSpaceStep as driven by ship_gym/game.py:185-195), compiled with
// -ffp-contract=off so every product and sum rounds exactly where the reference's does.
//
// Reference map (file:line under /root/reference):
//   ShipEnv.step                ship_gym/ship_env.py:136-156
//   handle_discrete_action      ship_gym/game.py:140-153      Ship.move_forward/rotate  ship_gym/models.py:129-146
//   LiDAR.query                 ship_gym/models.py:39-76
//   space.step -> cpSpaceStep   ship_gym/game.py:194          
//   collide_ship/collide_goal   ship_gym/game.py:232-257
//   determine_reward/is_done    ship_gym/ship_env.py:62-77,115-134
//   __add_states/closest_goal   ship_gym/ship_env.py:79-113, ship_gym/game.py:333-349
//   ShipEnv.reset/ShipGame.reset ship_gym/ship_env.py:171-184, ship_gym/game.py:260-277
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "shipsim.h"
#include "shipsim_internal.h"

namespace ssg {

// ---------------------------------------------------------------------------------------------------------
// map-record accessors: LDS-staged bank or per-lane global gathers
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double *lds_bank()
{
    extern __shared__ double s_bank[];
    return s_bank;
}

struct LdsRec {
    int off;
    __device__ __forceinline__ double operator[](int i) const { return lds_bank()[off + i]; }
};
struct GlbRec {
    const double *p;
    __device__ __forceinline__ double operator[](int i) const { return p[i]; }
};

__device__ __forceinline__ double dmin(double a, double b) { return (a < b) ? a : b; } // cpfmin
__device__ __forceinline__ double dmax(double a, double b) { return (a > b) ? a : b; } // cpfmax

// Stage `bytes` (multiple of 16) from global memory into LDS at offset 0 with LDS-DMA (global_load_lds_dwordx4:
// 1 KiB per wave-instruction, no VGPR round trip, all requests in flight at once), tail < 1 KiB through registers.
template <int BLOCK>
__device__ __forceinline__ void stage_bank_lds(const double *__restrict__ bank, int bytes)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NW = BLOCK / 64;
    const int nchunk = bytes >> 10;
    const char *g = reinterpret_cast<const char *>(bank);
    char *l = reinterpret_cast<char *>(lds_bank());
    for (int c = wave; c < nchunk; c += NW) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(l + c * 1024), 16, 0, 0);
    }
    const int tail0 = nchunk << 10;
    const int o = tail0 + (int)threadIdx.x * 16;
    if (o < bytes) *reinterpret_cast<double2 *>(l + o) = *reinterpret_cast<const double2 *>(g + o);
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) // src_lane must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// pick element k (0..4, per lane) of a 5-entry table held in wave-uniform registers
__device__ __forceinline__ double pick5(const double *t, int stride, int k)
{
    double r = t[0];
    r = (k == 1) ? t[stride] : r;
    r = (k == 2) ? t[2 * stride] : r;
    r = (k == 3) ? t[3 * stride] : r;
    r = (k == 4) ? t[4 * stride] : r;
    return r;
}

// ShipGame.closest_goal (game.py:333-349): strict '<', first listed goal wins ties; (-1,-1) when none left.
template <class Rec>
__device__ __forceinline__ void nearest_goal(const Rec &rec, unsigned gm, int n_goals, double x, double y, double &gx,
                                             double &gy)
{
    gx = -1.0;
    gy = -1.0;
    double best = INFINITY;
    for (int g = 0; g < n_goals; ++g) {
        const double px = rec[SSG_MAP_OFF_GOALS + 2 * g], py = rec[SSG_MAP_OFF_GOALS + 2 * g + 1];
        const double dx = px - x, dy = py - y;
        const double d = dx * dx + dy * dy; // squared distance orders exactly like Vec2d.get_distance's sqrt
        const bool take = (gm & (1u << g)) && (d < best); // first alive goal always beats +inf
        best = take ? d : best;
        gx = take ? px : gx;
        gy = take ? py : gy;
    }
}

// ---------------------------------------------------------------------------------------------------------
// LiDAR (models.py:39-76), wave-compacted.
//
// A beam can only touch a bank hull if the beam's bounding box meets the hull's (most beams of most ships do
// not: the river is wider than the 100-unit range).  Each lane therefore only CULLS its NB x 2 (beam, hull)
// pairs; the surviving pairs of the whole wave are compacted into a per-wave LDS queue (ballot + mbcnt) and
// processed 64 at a time, one pair per lane, so the plane loops run on dense wavefronts.  A worker lane pulls
// the pose of the env it serves with ds_bpermute, rebuilds the beam (same expressions as the owner lane) and runs
// cpShapeSegmentQuery(shape, a=(cx,cy), b=(ex,ey), r=0) against one hull:
//   EXACT = true : cpPolyShapeSegmentQuery literally — every plane is intersected (one division per plane),
//                  accepted when the crossing lies inside the edge's extent, later planes overwrite.
//   EXACT = false: the same predicate with one division per beam: among the planes the beam crosses front-to-back
//                  within its length (d >= 0 and d <= den, i.e. 0 <= t <= 1) only the one with the largest t can
//                  be the entry edge of a convex polygon, so only that plane gets the exact t = d/den, lerp and
//                  edge-extent test.  Identical results except when a ray passes within rounding of a hull vertex.
// Results travel back through two per-wave LDS arrays res0/res1[beam][lane] (-1 = no hit), one per hull; the owner
// lane then applies "the first shape in list order that reports a hit wins" (models.py:61-72): left bank first.
// ---------------------------------------------------------------------------------------------------------
// per-wave LDS scratch: res0[NB][64] + res1[NB][64] doubles, queue[2*NB*64 + 64 trash] u16, item counter (16 B)
__host__ __device__ __forceinline__ constexpr int lds_scratch_wave_bytes(int nb)
{
    return 2 * nb * 64 * 8 + (2 * nb * 64 + 64) * 2 + 16;
}

constexpr int kPlaneChunk = 4; // hull planes fetched from LDS ahead of their arithmetic, per loop trip

template <int NB, bool LDS_BANK, bool EXACT>
__device__ __forceinline__ void lidar_pass(const DevCfg &c, const int n_items, const unsigned short *queue, double *res0,
                                           double *res1, const double *beamtab, const double cx, const double cy,
                                           const double ca, const double sa, const int rec_off, const int lane)
{
    const auto bk = [&](int i) {
        if constexpr (LDS_BANK) return lds_bank()[i];
        else return c.bank[i];
    };
    for (int base = 0; base < n_items; base += 64) {
        const int idx = base + lane;
        const bool act = idx < n_items;
        const unsigned code = queue[act ? idx : 0];
        const int src = code & 63, bi = (code >> 6) & (SSG_MAX_BEAMS - 1), s = (code >> 10) & 1;
        const double wcx = __shfl(cx, src), wcy = __shfl(cy, src), wca = __shfl(ca, src), wsa = __shfl(sa, src);
        const int woff = __shfl(rec_off, src);
        const double cphi = beamtab[bi], sphi = beamtab[SSG_MAX_BEAMS + bi];
        const double ux = wca * cphi - wsa * sphi, uy = wsa * cphi + wca * sphi;
        const double ex = wcx + c.lidar_dist * ux, ey = wcy + c.lidar_dist * uy;
        const int cnt = (int)bk(woff + SSG_MAP_OFF_COUNTS + s);
        const int pb = woff + SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES);
        bool outside = false; // cpPolyShapePointQuery(a): some plane has a strictly in front
        bool ok = false;
        double ptx = ex, pty = ey;
        double bd = -1.0, bden = 1.0;
        int bj = 0;
        for (int j0 = 0; __any(act && (j0 < cnt)); j0 += kPlaneChunk) {
            double pv0x[kPlaneChunk], pv0y[kPlaneChunk], pnx[kPlaneChunk], pny[kPlaneChunk], pv0n[kPlaneChunk];
            double pdtmin[kPlaneChunk], pdtmax[kPlaneChunk];
#pragma unroll
            for (int u = 0; u < kPlaneChunk; ++u) { // all LDS reads of the chunk first: one latency per chunk
                const int j = j0 + u;
                const int q = pb + SSG_PLANE_DOUBLES * ((act && (j < cnt)) ? j : 0);
                pv0x[u] = bk(q + 0); pv0y[u] = bk(q + 1); pnx[u] = bk(q + 2); pny[u] = bk(q + 3); pv0n[u] = bk(q + 4);
                if (EXACT) { pdtmin[u] = bk(q + 5); pdtmax[u] = bk(q + 6); }
            }
#pragma unroll
            for (int u = 0; u < kPlaneChunk; ++u) {
                const int j = j0 + u;
                const bool valid = act && (j < cnt);
                const double v0x = pv0x[u], v0y = pv0y[u], nx = pnx[u], ny = pny[u], v0n = pv0n[u];
                outside = outside || (valid && ((nx * (wcx - v0x) + ny * (wcy - v0y)) > 0.0));
                const double an = wcx * nx + wcy * ny;
                const double d = an - v0n;
                const bool front = valid && !(d < 0.0);
                const double bn = ex * nx + ey * ny;
                const double den = dmax(an - bn, DBL_MIN);
                if (EXACT) {
                    const double t = d / den;
                    const double omt = 1.0 - t;
                    const double qx = wcx * omt + ex * t, qy = wcy * omt + ey * t; // cpvlerp(a,b,t)
                    const double dtv = nx * qy - ny * qx;                           // cpvcross(n, point)
                    const bool acc = front && !(t < 0.0 || 1.0 < t) && (pdtmin[u] <= dtv) && (dtv <= pdtmax[u]);
                    ok = ok || acc;
                    ptx = acc ? qx : ptx;
                    pty = acc ? qy : pty;
                } else {
                    // candidate: 0 <= d/den <= 1; better: d/den >= best (cross-multiplied, dens > 0; ties -> later)
                    const bool better = front && (d <= den) && (d * bden >= bd * den);
                    bd = better ? d : bd;
                    bden = better ? den : bden;
                    bj = better ? j : bj;
                }
            }
        }
        if (!EXACT) {
            const int q = pb + SSG_PLANE_DOUBLES * bj;
            const double nx = bk(q + 2), ny = bk(q + 3), dtmin = bk(q + 5), dtmax = bk(q + 6);
            const double t = bd / bden;
            const double omt = 1.0 - t;
            ptx = wcx * omt + ex * t;
            pty = wcy * omt + ey * t;
            const double dtv = nx * pty - ny * ptx;
            ok = (bd >= 0.0) && (dtmin <= dtv) && (dtv <= dtmax);
        }
        // start point inside (or on) the polygon: hit at alpha 0 whose reported point is the FAR end b (App. A.7)
        const bool hit = act && (outside ? ok : true);
        const double px = outside ? ptx : ex, py = outside ? pty : ey;
        const double dx = px - wcx, dy = py - wcy;
        const double dist = sqrt(dx * dx + dy * dy); // Vec2d.get_distance
        if (hit) (s ? res1 : res0)[bi * 64 + src] = dist;
    }
}

// Diagnostic stamps (-DSSG_STAMPS builds only; the product kernel executes none): s_memtime at section boundaries,
// written by lane 0 of every wave to a buffer nothing else reads.
#ifdef SSG_STAMPS
#define SSG_STAMP(k)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        stamp_[k] = t_;                                                                           \
    } while (0)
#else
#define SSG_STAMP(k) do { } while (0)
#endif

#ifdef SSG_ABLATION
#define SSG_ABL(bit) (c.flags & (1u << (16 + (bit))))
#else
#define SSG_ABL(bit) false
#endif

// ---------------------------------------------------------------------------------------------------------
// The step kernel
// ---------------------------------------------------------------------------------------------------------
template <int NB, int BLOCK, bool LDS_BANK, bool EXACT>
__global__ __launch_bounds__(BLOCK) void step_kernel(const DevCfg c, const int32_t *__restrict__ actions,
                                                     double *__restrict__ obs, double *__restrict__ reward_out,
                                                     uint8_t *__restrict__ done_out, uint8_t *__restrict__ flags_out)
{
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = e < c.n_envs;
    const size_t np = (size_t)c.n_pad;
    const int lane = threadIdx.x & 63;

    double *__restrict__ colX = c.f64cols + COL_X * np;
    double *__restrict__ colY = c.f64cols + COL_Y * np;
    double *__restrict__ colVX = c.f64cols + COL_VX * np;
    double *__restrict__ colVY = c.f64cols + COL_VY * np;
    double *__restrict__ colA = c.f64cols + COL_A * np;
    double *__restrict__ colW = c.f64cols + COL_W * np;
    double *__restrict__ colCum = c.f64cols + COL_CUM * np;
    double *__restrict__ colLid = c.f64cols + COL_LIDAR * np;
    int32_t *__restrict__ colRud = c.i32cols + ICOL_RUDDER * np;
    int32_t *__restrict__ colStep = c.i32cols + ICOL_STEP * np;
    int32_t *__restrict__ colMap = c.i32cols + ICOL_MAP * np;

#ifdef SSG_STAMPS
    unsigned long long stamp_[16] = {};
#endif
    SSG_STAMP(0);
    // ---- state loads first, then the bank DMA: everything below that does not need the bank (trigonometry, the
    //      action, the integrator, the ship's world transform) runs while the 100 KB of records stream into LDS.
    //      No early exit: lanes past n_envs stay active as workers of the wave-cooperative sections; they carry
    //      env 0's state, never count as "near" anything and store nothing. ----
    const int el = live ? e : 0;
    double x = colX[el], y = colY[el], vx = colVX[el], vy = colVY[el], ang = colA[el], w = colW[el];
    double cum = colCum[el];
    double lid[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) lid[i] = colLid[(size_t)i * np + el];
    int rudder = colRud[el], steps = colStep[el], map_id = colMap[el];
    unsigned gm = c.mask[el];
    const int act = actions[el];
    if (LDS_BANK && !SSG_ABL(8)) stage_bank_lds<BLOCK>(c.bank, c.n_maps * (SSG_MAP_STRIDE * 8));
    // LDS layout: [bank records (LDS_BANK only)] [beam cos/sin table 2 x 16 doubles] [per-wave lidar scratch]
    char *lds_scratch = reinterpret_cast<char *>(lds_bank()) + (LDS_BANK ? ((c.n_maps * (SSG_MAP_STRIDE * 8) + 15) & ~15) : 0);
    if (threadIdx.x < 2 * SSG_MAX_BEAMS)
        reinterpret_cast<double *>(lds_scratch)[threadIdx.x] =
            (threadIdx.x < SSG_MAX_BEAMS) ? c.beam_cos[threadIdx.x & (SSG_MAX_BEAMS - 1)] : c.beam_sin[threadIdx.x & (SSG_MAX_BEAMS - 1)];

    const int F = 6 + NB;
    const bool hist2 = c.history >= 2;
    const int rec_off = (int)map_id * SSG_MAP_STRIDE;

    // previous frame (oldest slot of the 2-frame history): a pure function of the pre-step state
    const double pf_x = x, pf_y = y, pf_rud = (double)rudder, pf_a = ang;

    // ---- handle_discrete_action (game.py:140-153) ----
    double sa0, ca0;
    sincos(ang, &sa0, &ca0); // cpvforangle(a) = (cos a, sin a): body->transform rotation
    double fx, fy, tq;
    {
        // Ship.move_forward -> cpBodyApplyForceAtLocalPoint(force_vector*1, point_of_thrust)
        const double px = (gm & 0x80u) ? (0.0 - (double)rudder) : c.px0; // models.py:109,146
        const double py = c.py0;
        const double fwx = (-sa0) * c.force_y, fwy = ca0 * c.force_y;    // cpTransformVect(transform, (0,F))
        const double pwx = ca0 * px + (-sa0) * py + x, pwy = sa0 * px + ca0 * py + y; // cpTransformPoint
        const double rx = pwx - x, ry = pwy - y;                         // minus transform * cog, cog = (0,0)
        const bool thrust = act == 0;
        fx = thrust ? fwx : 0.0;
        fy = thrust ? fwy : 0.0;
        tq = thrust ? (rx * fwy - ry * fwx) : 0.0;
    }
    if (act == 1 || act == 2) {
        // Ship.rotate(-5 / +5) + clamp_rudder (models.py:136-146)
        rudder += (act == 1) ? -c.rudder_step : c.rudder_step;
        rudder = max(-c.rudder_max, min(c.rudder_max, rudder));
        gm |= 0x80u;
    }

    // lidar origin on the PRE-step pose: pos + half the world AABB extents (models.py:51-53)
    double cx, cy;
    {
        double bl = INFINITY, br = -INFINITY, bb = INFINITY, bt = -INFINITY;
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const double hx = c.hull[2 * i], hy = c.hull[2 * i + 1];
            const double wx = ca0 * hx + (-sa0) * hy + x, wy = sa0 * hx + ca0 * hy + y;
            bl = dmin(bl, wx); br = dmax(br, wx);
            bb = dmin(bb, wy); bt = dmax(bt, wy);
        }
        cx = x + (br - bl) / 2;
        cy = y + (bt - bb) / 2;
    }
    const double x0 = x, y0 = y;

    // ---- cpSpaceStep (1): cpBodyUpdatePosition ----
    x = x + vx * c.dt;
    y = y + vy * c.dt;
    ang = ang + w * c.dt;
    double sa, ca;
    sincos(ang, &sa, &ca);

    // ---- cpSpaceStep (2): cpPolyShapeCacheData for the ship: world AABB (the planes are rebuilt by the workers) ----
    double sbl = INFINITY, sbr = -INFINITY, sbb = INFINITY, sbt = -INFINITY;
#pragma unroll
    for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
        const double hx = c.hull[2 * i], hy = c.hull[2 * i + 1];
        const double wx = ca * hx + (-sa) * hy + x, wy = sa * hx + ca * hy + y;
        sbl = dmin(sbl, wx); sbr = dmax(sbr, wx);
        sbb = dmin(sbb, wy); sbt = dmax(sbt, wy);
    }

    // ---- cpSpaceStep (3): cpBodyUpdateVelocity (gravity 0); forces are cleared afterwards.  (The narrowphase sits
    //      between (1) and (3) in Chipmunk but reads positions only, so the order here is immaterial.) ----
    vx = vx * c.damp + (fx * c.m_inv) * c.dt;
    vy = vy * c.damp + (fy * c.m_inv) * c.dt;
    w = w * c.damp + tq * c.i_inv * c.dt;
    // (4) impulse solver: its output cannot reach an observation before the env is reset.
    const bool oob_x = (x < 0.0) || (x > c.width);
    const bool oob_y = (y < 0.0) || (y > c.height);

    // worker coordinates of the wave-cooperative sections: lane L = 5*q + i
    const int wq = lane / 5, wi = lane - 5 * wq;
    const double w_hx = pick5(c.hull, 2, wi), w_hy = pick5(c.hull + 1, 2, wi);       // ship vertex i (local)
    const double w_nx = pick5(c.nrm, 2, wi), w_ny = pick5(c.nrm + 1, 2, wi);         // ship normal i (local)
    const int wip = (wi == 0) ? (SSG_SHIP_VERTS - 1) : (wi - 1);
    const double w_px = pick5(c.hull, 2, wip), w_py = pick5(c.hull + 1, 2, wip);     // previous vertex (edge start)

    SSG_STAMP(1);
    if (LDS_BANK) __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): the LDS-DMA writes of this wave have landed
    __syncthreads();
    SSG_STAMP(2);

    auto make_rec = [&](int m) {
        if constexpr (LDS_BANK) return LdsRec{m * SSG_MAP_STRIDE};
        else return GlbRec{c.bank + (size_t)m * SSG_MAP_STRIDE};
    };
    const auto rec = make_rec(map_id);

    double pf_gx = 0, pf_gy = 0;
    if (!SSG_ABL(0)) nearest_goal(rec, gm, c.n_goals, x0, y0, pf_gx, pf_gy);

    // ---- LiDAR.query on the PRE-step pose (models.py:39-76; game.py:193 runs it before space.step) ----
    double nl[NB]; // this step's new readings (-1 where nothing was hit)
#pragma unroll
    for (int i = 0; i < NB; ++i) nl[i] = -1.0;
    if (!SSG_ABL(1)) {
        // Beam i points along heading + phi_i, phi_i = rad(90 - spread/2) + i*rad(spread/n_beams) (models.py:48-49,
        // 62-64).  cos/sin(heading + phi_i) come from the body rotation and host-computed cos/sin(phi_i) by the
        // angle-addition identity instead of one sincos per beam: endpoints agree with the reference's to ~1e-13
        // (they only feed lidar readings, never the dynamics).
        unsigned need0 = 0, need1 = 0;
        {
            const double a0l = rec[SSG_MAP_OFF_AABB + 0], a0b = rec[SSG_MAP_OFF_AABB + 1];
            const double a0r = rec[SSG_MAP_OFF_AABB + 2], a0t = rec[SSG_MAP_OFF_AABB + 3];
            const double a1l = rec[SSG_MAP_OFF_AABB + 4], a1b = rec[SSG_MAP_OFF_AABB + 5];
            const double a1r = rec[SSG_MAP_OFF_AABB + 6], a1t = rec[SSG_MAP_OFF_AABB + 7];
            const double eps = 1e-6; // conservative margin: culling must never drop a pair the reference would hit
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const double ux = ca0 * c.beam_cos[i] - sa0 * c.beam_sin[i], uy = sa0 * c.beam_cos[i] + ca0 * c.beam_sin[i];
                const double ex = cx + c.lidar_dist * ux, ey = cy + c.lidar_dist * uy;
                const double lox = dmin(cx, ex) - eps, hix = dmax(cx, ex) + eps;
                const double loy = dmin(cy, ey) - eps, hiy = dmax(cy, ey) + eps;
                const bool n0 = live && (lox <= a0r) && (a0l <= hix) && (loy <= a0t) && (a0b <= hiy);
                const bool n1 = live && (lox <= a1r) && (a1l <= hix) && (loy <= a1t) && (a1b <= hiy);
                need0 |= n0 ? (1u << i) : 0u;
                need1 |= n1 ? (1u << i) : 0u;
            }
        }
        // per-wave LDS scratch
        char *wscr = lds_scratch + 2 * SSG_MAX_BEAMS * 8 + (threadIdx.x >> 6) * lds_scratch_wave_bytes(NB);
        double *res0 = reinterpret_cast<double *>(wscr);
        double *res1 = res0 + NB * 64;
        unsigned short *queue = reinterpret_cast<unsigned short *>(res1 + NB * 64);
        constexpr int kTrash = 2 * NB * 64; // 64 u16 past the queue swallow the writes of pairs that were culled
        int *counter = reinterpret_cast<int *>(queue + kTrash + 64);
        if (lane == 0) *counter = 0;
        const int mine = __popc(need0) + __popc(need1);
        int pos = (mine > 0) ? atomicAdd(counter, mine) : 0; // LDS atomic: compaction offset of this lane's pairs
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            res0[i * 64 + lane] = -1.0;
            res1[i * 64 + lane] = -1.0;
            const bool k0 = (need0 >> i) & 1u, k1 = (need1 >> i) & 1u;
            queue[k0 ? pos : (kTrash + lane)] = (unsigned short)(lane | (i << 6));
            pos += k0 ? 1 : 0;
            queue[k1 ? pos : (kTrash + lane)] = (unsigned short)(lane | (i << 6) | (1 << 10));
            pos += k1 ? 1 : 0;
        }
        const int n_items = __builtin_amdgcn_readfirstlane(*counter);
        SSG_STAMP(3);
        if (!SSG_ABL(3)) {
            const double *beamtab = reinterpret_cast<const double *>(lds_scratch);
            lidar_pass<NB, LDS_BANK, EXACT>(c, n_items, queue, res0, res1, beamtab, cx, cy, ca0, sa0, rec_off, lane);
        }
        SSG_STAMP(4);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            // first shape in list order that reports a hit wins (models.py:61-72): the left bank before the right
            const double r0 = res0[i * 64 + lane], r1 = res1[i * 64 + lane];
            nl[i] = (r0 >= 0.0) ? r0 : r1;
        }
    }
    SSG_STAMP(5);

    // ---- narrowphase, wave-cooperative ----------------------------------------------------------------------
    // Per lane only the cheap cpBBIntersects rejects run.  The few lanes that pass are then served one at a time
    // by the WHOLE wave: lane L = 5*q + i works on (bank plane q or goal q, ship vertex/edge i) of the served
    // env, whose pose is broadcast with v_readlane.  The arithmetic of every product and sum is exactly the
    // per-env formulation's (cpPolyShapeCacheData, SAT dot products, cpPolyShapePointQuery); only the min/any
    // reductions over vertices and planes are done with ballots instead of sequential loops.

    // player <-> bank hulls: collide_ship (game.py:232-241).  cpBBIntersects reject, then "closed convex sets
    // intersect" (GJK distance <= 0) evaluated as SAT over both polygons' edge normals: separated iff some axis
    // has every vertex of the other polygon strictly in front.
    bool colliding = false;
    {
        unsigned nearbits = 0;
        for (int s = 0; s < (SSG_ABL(4) ? 0 : 2); ++s) {
            const double al = rec[SSG_MAP_OFF_AABB + 4 * s + 0], ab = rec[SSG_MAP_OFF_AABB + 4 * s + 1];
            const double ar = rec[SSG_MAP_OFF_AABB + 4 * s + 2], at = rec[SSG_MAP_OFF_AABB + 4 * s + 3];
            const bool near = live && (sbl <= ar) && (al <= sbr) && (sbb <= at) && (ab <= sbt);
            nearbits |= near ? (1u << s) : 0u;
        }
        unsigned long long todo = __ballot(nearbits != 0u);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const double bx = readlane_f64(x, src), by = readlane_f64(y, src);
            const double bca = readlane_f64(ca, src), bsa = readlane_f64(sa, src);
            const int boff = __builtin_amdgcn_readlane(rec_off, src);
            const unsigned bnear = (unsigned)__builtin_amdgcn_readlane((int)nearbits, src);
            const double svx = bca * w_hx + (-bsa) * w_hy + bx, svy = bsa * w_hx + bca * w_hy + by;
            const double snx_ = bca * w_nx + (-bsa) * w_ny, sny_ = bsa * w_nx + bca * w_ny;
            const double off_i = snx_ * svx + sny_ * svy;
            bool col = false;
            for (int s = 0; s < 2; ++s) {
                if (!(bnear & (1u << s))) continue; // wave-uniform
                const auto brec = [&](int i) {
                    if constexpr (LDS_BANK) return lds_bank()[boff + i];
                    else return c.bank[(size_t)boff + i];
                };
                const int cnt = (int)brec(SSG_MAP_OFF_COUNTS + s);
                const bool valid = (lane < 60) && (wq < cnt);
                const int q = SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES) + SSG_PLANE_DOUBLES * (valid ? wq : 0);
                const double v0x = brec(q + 0), v0y = brec(q + 1), nx = brec(q + 2), ny = brec(q + 3), v0n = brec(q + 4);
                const bool frontA = (nx * svx + ny * svy) > v0n;           // ship vertex i in front of bank plane q
                const bool frontB = (snx_ * v0x + sny_ * v0y) > off_i;     // bank vertex q in front of ship plane i
                const unsigned long long mV = __ballot(valid);
                const unsigned long long mA = __ballot(valid && frontA);
                const unsigned long long missB = mV & ~__ballot(valid && frontB);
                const unsigned long long P = 0x0084210842108421ull;       // bit 5q, q = 0..11
                // axis = bank plane q: all five (q,i) bits set
                const unsigned long long allA = mA & (mA >> 1) & (mA >> 2) & (mA >> 3) & (mA >> 4) & P;
                // axis = ship plane i: no valid (q,i) bit missing
                bool sepB = false;
#pragma unroll
                for (int i = 0; i < SSG_SHIP_VERTS; ++i) sepB = sepB || (((missB >> i) & P) == 0ull);
                const bool separated = (allA != 0ull) || sepB;
                col = col || !separated;
            }
            colliding = (lane == src) ? col : colliding;
        }
    }
    SSG_STAMP(6);

    // player <-> goal circles: collide_goal (game.py:243-257).  Contact iff cpPolyShapePointQuery distance of the
    // centre to the ship hull <= radius (negative inside), after the cpBBIntersects reject.
    bool goal_reached = false;
    {
        unsigned nearmask = 0;
        for (int g = 0; g < (SSG_ABL(5) ? 0 : c.n_goals); ++g) {
            const double gx = rec[SSG_MAP_OFF_GOALS + 2 * g], gy = rec[SSG_MAP_OFF_GOALS + 2 * g + 1];
            const double r = c.goal_r;
            const bool near = live && (gm & (1u << g)) && ((gx - r) <= sbr) && (sbl <= (gx + r)) && ((gy - r) <= sbt) &&
                              (sbb <= (gy + r));
            nearmask |= near ? (1u << g) : 0u;
        }
        unsigned long long todo = __ballot(nearmask != 0u);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const double bx = readlane_f64(x, src), by = readlane_f64(y, src);
            const double bca = readlane_f64(ca, src), bsa = readlane_f64(sa, src);
            const int boff = __builtin_amdgcn_readlane(rec_off, src);
            const unsigned bnear = (unsigned)__builtin_amdgcn_readlane((int)nearmask, src);
            const auto brec = [&](int i) {
                if constexpr (LDS_BANK) return lds_bank()[boff + i];
                else return c.bank[(size_t)boff + i];
            };
            // lane (q = goal, i = ship edge from vertex i-1 to vertex i)
            const bool valid = (wq < SSG_MAX_GOALS - 1) && (bnear & (1u << wq));
            const int gq = valid ? wq : 0;
            const double gx = brec(SSG_MAP_OFF_GOALS + 2 * gq), gy = brec(SSG_MAP_OFF_GOALS + 2 * gq + 1);
            const double v1x = bca * w_hx + (-bsa) * w_hy + bx, v1y = bsa * w_hx + bca * w_hy + by;
            const double v0x = bca * w_px + (-bsa) * w_py + bx, v0y = bsa * w_px + bca * w_py + by;
            const double snx_ = bca * w_nx + (-bsa) * w_ny, sny_ = bsa * w_nx + bca * w_ny;
            const bool out_i = (snx_ * (gx - v1x) + sny_ * (gy - v1y)) > 0.0;
            // cpClosetPointOnSegment(p, v0, v1)
            const double dx = v0x - v1x, dy = v0y - v1y;
            const double tt = dmax(0.0, dmin((dx * (gx - v1x) + dy * (gy - v1y)) / (dx * dx + dy * dy), 1.0));
            const double qx = v1x + dx * tt, qy = v1y + dy * tt;
            const double ex_ = gx - qx, ey_ = gy - qy;
            const double dist = sqrt(ex_ * ex_ + ey_ * ey_);
            // min over the five edges of this goal (lanes 5q .. 5q+4), any(outside) over the same five lanes
            double md = dist;
#pragma unroll
            for (int k = 1; k < SSG_SHIP_VERTS; ++k) {
                int o = wi + k;
                o = (o >= SSG_SHIP_VERTS) ? o - SSG_SHIP_VERTS : o;
                md = dmin(md, __shfl(dist, 5 * wq + o));
            }
            const unsigned long long mo = __ballot(valid && out_i);
            const bool outside = ((mo >> (5 * wq)) & 31ull) != 0ull;
            const double sd = outside ? md : -md;
            const unsigned long long got = __ballot(valid && (wi == 0) && (sd <= c.goal_r)); // bit 5q = goal q consumed
            unsigned gotmask = 0;
#pragma unroll
            for (int g = 0; g < SSG_MAX_GOALS - 1; ++g) gotmask |= ((got >> (5 * g)) & 1ull) ? (1u << g) : 0u;
            if (lane == src) {
                goal_reached = gotmask != 0u;
                gm &= ~gotmask;
            }
        }
    }
    SSG_STAMP(7);

    // ---- determine_reward (ship_env.py:62-77) ----
    double rew = goal_reached ? 1.0 : ((oob_x || oob_y) ? -1.0 : -0.01);
    if ((c.flags & SSG_FLAG_FIX_COLLISION_REWARD) && colliding && !goal_reached) rew = -1.0;
    cum += rew;

    // ---- __add_states (ship_env.py:79-113) ----
    double nf_gx = 0, nf_gy = 0;
    if (!SSG_ABL(0)) nearest_goal(rec, gm, c.n_goals, x, y, nf_gx, nf_gy);

    // ---- step_count += 1; is_done (ship_env.py:115-134,152-154) ----
    steps += 1;
    const int steps_after = steps;
    const unsigned alive = gm & ((1u << c.n_goals) - 1u);
    const bool done = colliding || (alive == 0u) || oob_x || oob_y || (steps >= c.max_steps);
    const bool do_reset = done && (c.flags & SSG_FLAG_AUTO_RESET);

    if (live && !SSG_ABL(6)) {
        // Episode statistics, per handle.  Integer counters in kStatsSlots slots (slot = workgroup mod slots): no
        // single hot address, and integer adds commute, so the totals are bitwise reproducible run to run.
        // cum is a sum of {1, -1, -0.01} terms, so round(100*cum) is the exact return in hundredths.
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(c.stats) + 4 * (blockIdx.x % kStatsSlots);
        if (done) {
            atomicAdd(slot + 0, (unsigned long long)(long long)llrint(cum * 100.0));
            atomicAdd(slot + 1, (unsigned long long)steps);
            atomicAdd(slot + 2, 1ull);
        }
        if (goal_reached) atomicAdd(slot + 3, 1ull);
    }

    // observation values: the stepped frames, or (VecEnv auto-reset) ShipGame.reset + ShipEnv.reset onto the next
    // bank record: history of -1 then the spawn frame.
    double o_old[6 + NB], o_new[6 + NB];
    o_old[0] = pf_x; o_old[1] = pf_y; o_old[2] = pf_rud; o_old[3] = pf_a; o_old[4] = pf_gx; o_old[5] = pf_gy;
    o_new[0] = x; o_new[1] = y; o_new[2] = (double)rudder; o_new[3] = ang; o_new[4] = nf_gx; o_new[5] = nf_gy;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        o_old[6 + i] = lid[i];                            // readings before this step's query
        lid[i] = (nl[i] >= 0.0) ? nl[i] : lid[i];         // a miss keeps the previous reading (sticky, App. B-3)
        o_new[6 + i] = lid[i];
    }
    if (do_reset) {
        map_id = map_id + 1;
        if (map_id >= c.n_maps) map_id = 0;
        const auto nrec = make_rec(map_id);
        x = c.spawn_x; y = c.spawn_y; vx = 0.0; vy = 0.0; ang = 0.0; w = 0.0; cum = 0.0;
        rudder = 0; steps = 0;
        gm = (1u << c.n_goals) - 1u;
#pragma unroll
        for (int i = 0; i < NB; ++i) lid[i] = -1.0;
#pragma unroll
        for (int i = 0; i < 6 + NB; ++i) { o_old[i] = -1.0; o_new[i] = -1.0; }
        o_new[0] = x; o_new[1] = y; o_new[2] = 0.0; o_new[3] = 0.0;
        o_new[4] = nrec[SSG_MAP_OFF_SPAWN_GOAL]; o_new[5] = nrec[SSG_MAP_OFF_SPAWN_GOAL + 1];
    }
    SSG_STAMP(8);
#ifdef SSG_STAMPS
    if (c.dbg && lane == 0) {
        unsigned long long *d_ = c.dbg + 16 * (size_t)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
        for (int k = 0; k < 9; ++k) d_[k] = stamp_[k];
    }
#endif
    if (!live) return; // every cooperative section is behind us: lanes past n_envs store nothing
    double *__restrict__ orow = obs + (size_t)e * (size_t)(F * c.history);
    if (!SSG_ABL(7)) {
        if (hist2) {
#pragma unroll
            for (int i = 0; i < 6 + NB; ++i) orow[i] = o_old[i];
            orow += F;
        }
#pragma unroll
        for (int i = 0; i < 6 + NB; ++i) orow[i] = o_new[i];
    } else {
        double acc = 0;
        for (int i = 0; i < 6 + NB; ++i) acc += o_old[i] + o_new[i];
        orow[0] = acc;
    }

    reward_out[e] = rew;
    done_out[e] = done ? 1 : 0;
    if (flags_out) {
        unsigned ev = 0;
        if (colliding) ev |= SSG_EV_COLLIDING;
        if (goal_reached) ev |= SSG_EV_GOAL_REACHED;
        if (oob_x || oob_y) ev |= SSG_EV_OUT_OF_BOUNDS;
        if (steps_after >= c.max_steps) ev |= SSG_EV_MAX_STEPS;
        if (alive == 0u) ev |= SSG_EV_NO_GOALS_LEFT;
        flags_out[e] = (uint8_t)ev;
    }

    colX[e] = x; colY[e] = y; colVX[e] = vx; colVY[e] = vy; colA[e] = ang; colW[e] = w; colCum[e] = cum;
#pragma unroll
    for (int i = 0; i < NB; ++i) colLid[(size_t)i * np + e] = lid[i];
    colRud[e] = rudder; colStep[e] = steps; colMap[e] = map_id;
    c.mask[e] = (uint8_t)gm;
}

// ---------------------------------------------------------------------------------------------------------
// reset kernel: ShipEnv.reset / ShipGame.reset for the masked envs (ship_env.py:171-184, game.py:260-277)
// ---------------------------------------------------------------------------------------------------------
__global__ void reset_kernel(const DevCfg c, const uint8_t *__restrict__ mask, const int32_t *__restrict__ map_ids,
                             double *__restrict__ obs)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    if (mask && !mask[e]) return;
    const size_t np = (size_t)c.n_pad;
    int m;
    if (map_ids) m = map_ids[e];
    else m = (int)((c.env_id_base + (long long)e) % (long long)c.n_maps);
    const double *rec = c.bank + (size_t)m * SSG_MAP_STRIDE;
    c.f64cols[COL_X * np + e] = c.spawn_x;
    c.f64cols[COL_Y * np + e] = c.spawn_y;
    c.f64cols[COL_VX * np + e] = 0.0;
    c.f64cols[COL_VY * np + e] = 0.0;
    c.f64cols[COL_A * np + e] = 0.0;
    c.f64cols[COL_W * np + e] = 0.0;
    c.f64cols[COL_CUM * np + e] = 0.0;
    for (int i = 0; i < c.n_beams; ++i) c.f64cols[(COL_LIDAR + i) * np + e] = -1.0;
    c.i32cols[ICOL_RUDDER * np + e] = 0;
    c.i32cols[ICOL_STEP * np + e] = 0;
    c.i32cols[ICOL_MAP * np + e] = m;
    c.mask[e] = (uint8_t)((1u << c.n_goals) - 1u);
    if (obs) {
        const int F = 6 + c.n_beams;
        double *orow = obs + (size_t)e * (size_t)(F * c.history);
        for (int i = 0; i < F * (c.history - 1); ++i) orow[i] = -1.0; // deque([-1]*n), ship_env.py:180-181
        orow += F * (c.history - 1);
        orow[0] = c.spawn_x; orow[1] = c.spawn_y; orow[2] = 0.0; orow[3] = 0.0;
        orow[4] = rec[SSG_MAP_OFF_SPAWN_GOAL]; orow[5] = rec[SSG_MAP_OFF_SPAWN_GOAL + 1];
        for (int i = 0; i < c.n_beams; ++i) orow[6 + i] = -1.0;
    }
}

// ---------------------------------------------------------------------------------------------------------
// counter-based action stream: Philox4x32-10, counter = (env_lo, env_hi, step_lo, step_hi), key = seed
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&ctr)[4], const uint32_t (&key)[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ key[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ key[1];
    const uint32_t n3 = (uint32_t)p0;
    ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
}

__global__ void fill_actions_kernel(uint64_t seed, uint64_t step0, int K, long long env_base, int n,
                                    int32_t *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)K * n;
    if (idx >= total) return;
    const int k = (int)(idx / n), e = (int)(idx % n);
    const uint64_t env = (uint64_t)(env_base + e), step = step0 + (uint64_t)k;
    uint32_t ctr[4] = {(uint32_t)env, (uint32_t)(env >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { key[0] += 0x9E3779B9u; key[1] += 0xBB67AE85u; }
        philox_round(ctr, key);
    }
    out[idx] = (int32_t)(((uint64_t)ctr[0] * 3u) >> 32);
}

// ---------------------------------------------------------------------------------------------------------
// launchers (called from shipsim_api.cpp)
// ---------------------------------------------------------------------------------------------------------
using step_fn_t = void (*)(const DevCfg, const int32_t *, double *, double *, uint8_t *, uint8_t *);

template <int NB, int BLOCK>
static step_fn_t step_fn_nb(bool lds, bool exact)
{
    if (exact) return lds ? step_kernel<NB, BLOCK, true, true> : step_kernel<NB, BLOCK, false, true>;
    return lds ? step_kernel<NB, BLOCK, true, false> : step_kernel<NB, BLOCK, false, false>;
}

template <int BLOCK>
static step_fn_t step_fn_block(int nb, bool lds, bool exact)
{
    switch (nb) {
#define SSG_CASE(NB_) \
    case NB_: return step_fn_nb<NB_, BLOCK>(lds, exact);
#ifdef SSG_DEV_BUILD /* development builds instantiate the two BASELINE beam counts only */
        SSG_CASE(8) SSG_CASE(10)
#else
        SSG_CASE(1) SSG_CASE(2) SSG_CASE(3) SSG_CASE(4) SSG_CASE(5) SSG_CASE(6) SSG_CASE(7) SSG_CASE(8)
        SSG_CASE(9) SSG_CASE(10) SSG_CASE(11) SSG_CASE(12) SSG_CASE(13) SSG_CASE(14) SSG_CASE(15) SSG_CASE(16)
#endif
#undef SSG_CASE
    default: return nullptr;
    }
}

static step_fn_t step_fn(int nb, int block, bool lds, bool exact)
{
    switch (block) {
    case 64: return step_fn_block<64>(nb, lds, exact);
    case 256: return step_fn_block<256>(nb, lds, exact);
    case 512: return step_fn_block<512>(nb, lds, exact);
    default: return nullptr;
    }
}

// dynamic LDS: [bank (if staged)] [beam table] [per-wave lidar scratch]
size_t step_lds_bytes(int n_beams, int block, bool lds_bank, int n_maps)
{
    size_t b = lds_bank ? (((size_t)n_maps * SSG_MAP_STRIDE * 8 + 15) & ~(size_t)15) : 0;
    b += 2 * SSG_MAX_BEAMS * 8;
    b += (size_t)(block / 64) * (size_t)lds_scratch_wave_bytes(n_beams);
    return b;
}

// Raise the dynamic-LDS cap of the selected instantiation once (whenever the bank size changes).
hipError_t prepare_step(const DevCfg &c, int block, bool lds, size_t lds_bytes)
{
    step_fn_t k = step_fn(c.n_beams, block, lds, (c.flags & SSG_FLAG_EXACT_LIDAR) != 0);
    if (!k) return hipErrorInvalidValue;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_bytes);
}

hipError_t launch_step(const DevCfg &c, int block, bool lds, size_t lds_bytes, const int32_t *actions, double *obs,
                       double *reward, uint8_t *done, uint8_t *flags, hipStream_t stream)
{
    step_fn_t k = step_fn(c.n_beams, block, lds, (c.flags & SSG_FLAG_EXACT_LIDAR) != 0);
    if (!k) return hipErrorInvalidValue;
    const int grid = (c.n_envs + block - 1) / block;
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds_bytes, stream, c, actions, obs, reward, done, flags);
    return hipGetLastError();
}

hipError_t launch_reset(const DevCfg &c, const uint8_t *mask, const int32_t *map_ids, double *obs, hipStream_t stream)
{
    const int block = 256, grid = (c.n_envs + block - 1) / block;
    hipLaunchKernelGGL(reset_kernel, dim3(grid), dim3(block), 0, stream, c, mask, map_ids, obs);
    return hipGetLastError();
}

hipError_t launch_fill_actions(uint64_t seed, uint64_t step0, int K, long long env_base, int n, int32_t *out,
                               hipStream_t stream)
{
    const long long total = (long long)K * n;
    const int block = 256;
    const long long grid = (total + block - 1) / block;
    hipLaunchKernelGGL(fill_actions_kernel, dim3((unsigned)grid), dim3(block), 0, stream, seed, step0, K, env_base, n,
                       out);
    return hipGetLastError();
}

} // namespace ssg
