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
        const double d = sqrt(dx * dx + dy * dy);
        const bool take = (gm & (1u << g)) && (d < best); // first alive goal always beats +inf
        best = take ? d : best;
        gx = take ? px : gx;
        gy = take ? py : gy;
    }
}

// One bank hull against the NB lidar beams of this lane: cpShapeSegmentQuery(shape, a=(cx,cy), b=(ex,ey), r=0).
//   EXACT = true : cpPolyShapeSegmentQuery literally — every plane is intersected (one division per plane and
//                  beam), accepted when the crossing lies inside the edge's extent, later planes overwrite.
//   EXACT = false: the same predicate evaluated with one division per beam: among the planes the beam crosses
//                  front-to-back within its length (d >= 0 and d <= den, i.e. 0 <= t <= 1) only the one with the
//                  largest t can be the entry edge of a convex polygon, so only that plane gets the exact
//                  t = d/den, lerp and edge-extent test of the reference.  Identical results except when a ray
//                  passes within rounding of a hull vertex (then: adjacent edge, same point to ~1e-13).
// Branch-free over lanes; the trip count is the wave-wide maximum plane count.
template <int NB, bool EXACT, class Rec>
__device__ __forceinline__ void lidar_hull(const Rec &rec, int s, double cx, double cy, const double (&ex)[NB],
                                           const double (&ey)[NB], unsigned &hit, double (&hx)[NB], double (&hy)[NB])
{
    const int cnt = (int)rec[SSG_MAP_OFF_COUNTS + s];
    const int pbase = SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES);
    bool outside = false; // cpPolyShapePointQuery(a): any plane with a strictly in front
    hit = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) { hx[i] = ex[i]; hy[i] = ey[i]; }
    double bd[NB], bden[NB];
    int bj[NB];
    if (!EXACT) {
#pragma unroll
        for (int i = 0; i < NB; ++i) { bd[i] = -1.0; bden[i] = 1.0; bj[i] = 0; }
    }
    for (int j = 0; __any(j < cnt); ++j) {
        const bool valid = j < cnt;
        const int jj = valid ? j : 0;
        const double v0x = rec[pbase + 8 * jj + 0], v0y = rec[pbase + 8 * jj + 1];
        const double nx = rec[pbase + 8 * jj + 2], ny = rec[pbase + 8 * jj + 3];
        const double v0n = rec[pbase + 8 * jj + 4];
        outside = outside || (valid && ((nx * (cx - v0x) + ny * (cy - v0y)) > 0.0));
        const double an = cx * nx + cy * ny;
        const double d = an - v0n;
        const bool front = valid && !(d < 0.0);
        if (EXACT) {
            const double dtmin = rec[pbase + 8 * jj + 5], dtmax = rec[pbase + 8 * jj + 6];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const double bn = ex[i] * nx + ey[i] * ny;
                const double t = d / dmax(an - bn, DBL_MIN);
                const double omt = 1.0 - t;
                const double ptx = cx * omt + ex[i] * t, pty = cy * omt + ey[i] * t; // cpvlerp(a,b,t)
                const double dtv = nx * pty - ny * ptx;                               // cpvcross(n, point)
                const bool ok = front && !(t < 0.0 || 1.0 < t) && (dtmin <= dtv) && (dtv <= dtmax);
                hit |= ok ? (1u << i) : 0u;
                hx[i] = ok ? ptx : hx[i];
                hy[i] = ok ? pty : hy[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const double bn = ex[i] * nx + ey[i] * ny;
                const double den = dmax(an - bn, DBL_MIN);
                // candidate: 0 <= d/den <= 1; better: d/den >= best (cross-multiplied, dens > 0; ties -> later plane)
                const bool better = front && (d <= den) && (d * bden[i] >= bd[i] * den);
                bd[i] = better ? d : bd[i];
                bden[i] = better ? den : bden[i];
                bj[i] = better ? j : bj[i];
            }
        }
    }
    if (!EXACT) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int q = pbase + 8 * bj[i];
            const double nx = rec[q + 2], ny = rec[q + 3], dtmin = rec[q + 5], dtmax = rec[q + 6];
            const double t = bd[i] / bden[i];
            const double omt = 1.0 - t;
            const double ptx = cx * omt + ex[i] * t, pty = cy * omt + ey[i] * t;
            const double dtv = nx * pty - ny * ptx;
            const bool ok = (bd[i] >= 0.0) && (dtmin <= dtv) && (dtv <= dtmax);
            hit |= ok ? (1u << i) : 0u;
            hx[i] = ptx;
            hy[i] = pty;
        }
    }
    // start point inside (or on) the polygon: hit at alpha 0 whose reported point is the FAR end b (App. A.7)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        hx[i] = outside ? hx[i] : ex[i];
        hy[i] = outside ? hy[i] : ey[i];
    }
    hit = outside ? hit : ((1u << NB) - 1u);
}

// Timing-only ablation switches (development builds with -DSSG_ABLATION; never in the product library): bits
// 16.. of DevCfg.flags skip a section so its share of the kernel time can be measured.  Outputs are wrong.
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

    // ---- the bank DMA and every state load are issued back to back; one wait covers them all ----
    if (LDS_BANK && !SSG_ABL(8)) stage_bank_lds<BLOCK>(c.bank, c.n_maps * (SSG_MAP_STRIDE * 8));
    const int el = live ? e : 0;
    double x = colX[el], y = colY[el], vx = colVX[el], vy = colVY[el], ang = colA[el], w = colW[el];
    double cum = colCum[el];
    double lid[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) lid[i] = colLid[(size_t)i * np + el];
    int rudder = colRud[el], steps = colStep[el], map_id = colMap[el];
    unsigned gm = c.mask[el];
    const int act = actions[el];
    if (LDS_BANK) {
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0): the LDS-DMA writes of this wave have landed
        __syncthreads();
    }
    if (!live) return;

    auto make_rec = [&](int m) {
        if constexpr (LDS_BANK) return LdsRec{m * SSG_MAP_STRIDE};
        else return GlbRec{c.bank + (size_t)m * SSG_MAP_STRIDE};
    };
    const auto rec = make_rec(map_id);

    const int F = 6 + NB;
    const bool hist2 = c.history >= 2;

    // ---- previous frame (oldest slot of the 2-frame history) is a pure function of the pre-step state ----
    const double pf_x = x, pf_y = y, pf_rud = (double)rudder, pf_a = ang;
    double pf_gx = 0, pf_gy = 0;
    if (!SSG_ABL(0)) nearest_goal(rec, gm, c.n_goals, x, y, pf_gx, pf_gy);
    double pf_lid[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) pf_lid[i] = lid[i];

    // ---- handle_discrete_action (game.py:140-153) ----
    double sa, ca;
    sincos(ang, &sa, &ca); // cpvforangle(a) = (cos a, sin a): body->transform rotation
    double fx, fy, tq;
    {
        // Ship.move_forward -> cpBodyApplyForceAtLocalPoint(force_vector*1, point_of_thrust)
        const double px = (gm & 0x80u) ? (0.0 - (double)rudder) : c.px0; // models.py:109,146
        const double py = c.py0;
        const double fwx = (-sa) * c.force_y, fwy = ca * c.force_y;      // cpTransformVect(transform, (0,F))
        const double pwx = ca * px + (-sa) * py + x, pwy = sa * px + ca * py + y; // cpTransformPoint
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

    // ---- LiDAR.query on the PRE-step pose (models.py:39-76; game.py:193 runs it before space.step) ----
    if (!SSG_ABL(1)) {
        double bl = INFINITY, br = -INFINITY, bb = INFINITY, bt = -INFINITY;
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const double hx = c.hull[2 * i], hy = c.hull[2 * i + 1];
            const double wx = ca * hx + (-sa) * hy + x, wy = sa * hx + ca * hy + y;
            bl = dmin(bl, wx); br = dmax(br, wx);
            bb = dmin(bb, wy); bt = dmax(bt, wy);
        }
        const double cx = x + (br - bl) / 2, cy = y + (bt - bb) / 2; // models.py:51-53: pos + half AABB extents
        const double deg2rad = 0.017453292519943295;                  // CPython math.radians: pi/180
        const double angle_delta = (c.spread_deg / (double)NB) * deg2rad;
        const double angle_start = ang + (90.0 - c.spread_deg / 2) * deg2rad;

        double ex[NB], ey[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const double rotation = angle_start + (angle_delta * (double)i);
            double sr = rotation, cr = 1.0 - rotation;
            if (!SSG_ABL(2)) sincos(rotation, &sr, &cr);
            ex[i] = cx + c.lidar_dist * cr;
            ey[i] = cy + c.lidar_dist * sr;
        }
        unsigned hit0 = 0, hit1 = 0;
        double h0x[NB], h0y[NB], h1x[NB], h1y[NB];
        if (!SSG_ABL(3)) {
            lidar_hull<NB, EXACT>(rec, 0, cx, cy, ex, ey, hit0, h0x, h0y);
            lidar_hull<NB, EXACT>(rec, 1, cx, cy, ex, ey, hit1, h1x, h1y);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) { h0x[i] = ex[i]; h0y[i] = ey[i]; h1x[i] = ey[i]; h1y[i] = ex[i]; }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            // first shape in list order that reports a hit wins (models.py:61-72); a miss keeps the old value
            const bool a0 = hit0 & (1u << i), a1 = hit1 & (1u << i);
            const double px = a0 ? h0x[i] : h1x[i], py = a0 ? h0y[i] : h1y[i];
            const double dx = px - cx, dy = py - cy;
            const double dist = sqrt(dx * dx + dy * dy); // Vec2d.get_distance
            lid[i] = (a0 || a1) ? dist : lid[i];
        }
    }

    // ---- cpSpaceStep (1): cpBodyUpdatePosition ----
    x = x + vx * c.dt;
    y = y + vy * c.dt;
    ang = ang + w * c.dt;
    sincos(ang, &sa, &ca);

    // ---- cpSpaceStep (2): cpPolyShapeCacheData for the ship, then the narrowphase ----
    double swx[SSG_SHIP_VERTS], swy[SSG_SHIP_VERTS], snx[SSG_SHIP_VERTS], sny[SSG_SHIP_VERTS];
    double sbl = INFINITY, sbr = -INFINITY, sbb = INFINITY, sbt = -INFINITY;
#pragma unroll
    for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
        const double hx = c.hull[2 * i], hy = c.hull[2 * i + 1];
        const double lnx = c.nrm[2 * i], lny = c.nrm[2 * i + 1];
        swx[i] = ca * hx + (-sa) * hy + x;
        swy[i] = sa * hx + ca * hy + y;
        snx[i] = ca * lnx + (-sa) * lny;
        sny[i] = sa * lnx + ca * lny;
        sbl = dmin(sbl, swx[i]); sbr = dmax(sbr, swx[i]);
        sbb = dmin(sbb, swy[i]); sbt = dmax(sbt, swy[i]);
    }

    // player <-> bank hulls: collide_ship (game.py:232-241).  cpBBIntersects reject, then "closed convex sets
    // intersect" (GJK distance <= 0) evaluated as SAT over both polygons' edge normals.  A hull is skipped only
    // when no lane of the wave passes the AABB test.
    bool colliding = false;
    for (int s = 0; s < (SSG_ABL(4) ? 0 : 2); ++s) {
        const double al = rec[SSG_MAP_OFF_AABB + 4 * s + 0], ab = rec[SSG_MAP_OFF_AABB + 4 * s + 1];
        const double ar = rec[SSG_MAP_OFF_AABB + 4 * s + 2], at = rec[SSG_MAP_OFF_AABB + 4 * s + 3];
        const bool near = (sbl <= ar) && (al <= sbr) && (sbb <= at) && (ab <= sbt);
        if (!__any(near)) continue;
        const int cnt = (int)rec[SSG_MAP_OFF_COUNTS + s];
        const int pbase = SSG_MAP_OFF_PLANES + s * (SSG_MAX_HULL * SSG_PLANE_DOUBLES);
        bool separated = false;
        double mn_ship_axis[SSG_SHIP_VERTS]; // min over bank verts of dot(ship normal i, v)
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) mn_ship_axis[i] = INFINITY;
        for (int j = 0; __any(j < cnt); ++j) {
            const bool valid = j < cnt;
            const int jj = valid ? j : 0;
            const double v0x = rec[pbase + 8 * jj + 0], v0y = rec[pbase + 8 * jj + 1];
            const double nx = rec[pbase + 8 * jj + 2], ny = rec[pbase + 8 * jj + 3];
            const double v0n = rec[pbase + 8 * jj + 4];
            double mn = INFINITY;
#pragma unroll
            for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
                mn = dmin(mn, nx * swx[i] + ny * swy[i]);
                mn_ship_axis[i] = dmin(mn_ship_axis[i], snx[i] * v0x + sny[i] * v0y); // jj=0 repeats a real vertex
            }
            separated = separated || (valid && (mn > v0n));
        }
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const double off = snx[i] * swx[i] + sny[i] * swy[i];
            separated = separated || (mn_ship_axis[i] > off);
        }
        colliding = colliding || (near && !separated);
    }

    // player <-> goal circles: collide_goal (game.py:243-257).  Contact iff cpPolyShapePointQuery distance of the
    // centre to the ship hull <= radius (negative inside), after the cpBBIntersects reject.
    bool goal_reached = false;
    for (int g = 0; g < (SSG_ABL(5) ? 0 : c.n_goals); ++g) {
        const double gx = rec[SSG_MAP_OFF_GOALS + 2 * g], gy = rec[SSG_MAP_OFF_GOALS + 2 * g + 1];
        const double r = c.goal_r;
        const bool near = (gm & (1u << g)) && ((gx - r) <= sbr) && (sbl <= (gx + r)) && ((gy - r) <= sbt) &&
                          (sbb <= (gy + r));
        if (!__any(near)) continue;
        bool outside = false;
        double min_dist = INFINITY;
        double v0x = swx[SSG_SHIP_VERTS - 1], v0y = swy[SSG_SHIP_VERTS - 1];
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const double v1x = swx[i], v1y = swy[i];
            outside = outside || ((snx[i] * (gx - v1x) + sny[i] * (gy - v1y)) > 0.0);
            // cpClosetPointOnSegment(p, v0, v1)
            const double dx = v0x - v1x, dy = v0y - v1y;
            const double tt = dmax(0.0, dmin((dx * (gx - v1x) + dy * (gy - v1y)) / (dx * dx + dy * dy), 1.0));
            const double qx = v1x + dx * tt, qy = v1y + dy * tt;
            const double ex_ = gx - qx, ey_ = gy - qy;
            const double dist = sqrt(ex_ * ex_ + ey_ * ey_);
            min_dist = (dist < min_dist) ? dist : min_dist;
            v0x = v1x;
            v0y = v1y;
        }
        const double sd = outside ? min_dist : -min_dist;
        const bool got = near && (sd <= r);
        goal_reached = goal_reached || got;
        gm = got ? (gm & ~(1u << g)) : gm;
    }

    // ---- cpSpaceStep (3): cpBodyUpdateVelocity (gravity 0); forces are cleared afterwards ----
    vx = vx * c.damp + (fx * c.m_inv) * c.dt;
    vy = vy * c.damp + (fy * c.m_inv) * c.dt;
    w = w * c.damp + tq * c.i_inv * c.dt;
    // (4) impulse solver: its output cannot reach an observation before the env is reset.

    // ---- determine_reward (ship_env.py:62-77) ----
    const bool oob_x = (x < 0.0) || (x > c.width);
    const bool oob_y = (y < 0.0) || (y > c.height);
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

    double *__restrict__ orow = obs + (size_t)e * (size_t)(F * c.history);
    const bool do_reset = done && (c.flags & SSG_FLAG_AUTO_RESET);

    if (!SSG_ABL(6)) {
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
    for (int i = 0; i < NB; ++i) { o_old[6 + i] = pf_lid[i]; o_new[6 + i] = lid[i]; }
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
    if (!SSG_ABL(7)) {
    if (hist2) {
#pragma unroll
        for (int i = 0; i < 6 + NB; ++i) orow[i] = o_old[i];
        orow += F;
    }
#pragma unroll
    for (int i = 0; i < 6 + NB; ++i) orow[i] = o_new[i];
    } else { double acc = 0; for (int i = 0; i < 6 + NB; ++i) acc += o_old[i] + o_new[i]; orow[0] = acc; }

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

// Raise the dynamic-LDS cap of the selected instantiation once (whenever the bank size changes).
hipError_t prepare_step(const DevCfg &c, int block, bool lds, size_t lds_bytes)
{
    step_fn_t k = step_fn(c.n_beams, block, lds, (c.flags & SSG_FLAG_EXACT_LIDAR) != 0);
    if (!k) return hipErrorInvalidValue;
    if (!lds) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_bytes);
}

hipError_t launch_step(const DevCfg &c, int block, bool lds, size_t lds_bytes, const int32_t *actions, double *obs,
                       double *reward, uint8_t *done, uint8_t *flags, hipStream_t stream)
{
    step_fn_t k = step_fn(c.n_beams, block, lds, (c.flags & SSG_FLAG_EXACT_LIDAR) != 0);
    if (!k) return hipErrorInvalidValue;
    const int grid = (c.n_envs + block - 1) / block;
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds ? lds_bytes : 0, stream, c, actions, obs, reward, done, flags);
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
