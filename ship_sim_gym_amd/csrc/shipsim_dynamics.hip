// shipsim_dynamics.hip — config 4 (BASELINE configs[3]: 4 ships per env): the bodies other than the player.
//
// Replaces, for N envs at once, what `space.step(dt)` (game.py:194 -> Chipmunk2D cpSpaceStep) does to the three
// traffic ships of ShipGame.add_default_traffic (game.py:279-286, add_ship game.py:117-131, Ship.__init__
// models.py:87-111) and to the five goal bodies (add_goal game.py:77-95: mass-1 dynamic circles), including the
// contact solver between them and the river banks:
//   cpBodyUpdatePosition (v_bias / w_bias), cpPolyShapeCacheData, cpCollide (GJK + EPA closest points, support-edge
//   clipping, circle cases), cpArbiterUpdate / PreStep / ApplyCachedImpulse / ApplyImpulse (10 iterations),
//   cpSpaceArbiterSetFilter (collision persistence 3), cpBodyUpdateVelocity.
// (The player's `collide_ship` begin-callback against traffic — collision_type 1, models.py:100; game.py:232-241 — is the step
// kernel's since round 4: it reads the poses, and the rotations of the angle columns, this file's kernels leave behind.)
//
// Design (MI355X).  One lane per env, one wave per workgroup.  The work per env is a short, branchy, strictly
// sequential chain (Gauss-Seidel over at most a handful of contacts), so the kernel is bound by the latency of that
// chain, not by HBM: what matters is that nothing on the chain goes to memory.
//  * Body state and the solver's arbiter records live in LDS, one column per field, lane-contiguous
//    (`lds[field*kGrp + lane]`, kGrp = 48 envs per wave): the per-lane gathers of the solver are (nearly) conflict-free.
//    (A first version kept them in per-lane arrays: 8.7 KB of scratch per lane, 1 GB of HBM traffic per step, 420 us.)
//  * Shapes are never materialised: a ship's world vertices are its pose applied on the fly to the hull constants
//    (staged once per workgroup in LDS, broadcast reads).  The bank planes: the queue is walked map-major and every map's
//    stretch starts on a wave boundary, so all lanes of a wave sit on ONE bank record, staged once per wave (96 doubles,
//    broadcast reads) — dyn_step_kernel<true>; banks of more than 64 records and per-env rings of worlds keep the planes in
//    per-lane columns (dyn_step_kernel<false>).  76 KB of LDS per wave: two waves per CU, each alone on its SIMD, 512 at
//    once = 24 576 envs in one round (64 envs per wave with per-lane planes was 158 KB: 256 waves, 16 384 envs, and a
//    second round — twice the time — whenever more than a quarter of the batch was in its post-reset transient).
//    EPA's growing hull has seven LDS entries per lane, scratch beyond (practically never).
//  * Arbiter records (accumulated impulses, contact hashes, state, age) persist in struct-of-arrays columns but are
//    read or written only for pairs whose bit is set in the env's 64-bit live mask.
//  * Which envs are stepped: the ones the step kernel's body role queued at the end of the previous step, each in the array of
//    its (bank record, steps since the reset) bucket (DevCfg::dyn_bucket) so that the lanes of a wave walk the same path; a wave
//    finds its entries from the 512 bucket counters alone (no sort pass) and reads the bodies through a row-major shadow of
//    their columns (DevCfg::dyn_row).  Envs whose space is at a fixed point of cpSpaceStep (the rest bit) are not stepped at all.
// Per step: dyn_step_kernel, then the step kernel, which reads this step's goal and traffic positions from the dyn columns
// (DevCfg::dyn_*) and queues the envs for the next step.
//
// The canonical pair order, the cold GJK start and the unsolved player arbiters are the named assumptions of the
// oracle (oracle/ssg_dynamics.c header); this file follows the same ones.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "shipsim_internal.h"

namespace ssg {
namespace {

extern __shared__ double lds[]; // [field][64 lanes] columns, then the wave-uniform hull constants

// Envs per workgroup of the full step (lanes kGrp..63 of its one wave idle: a lone wave's FP64 chain takes the same time
// whatever its width).  Measured at 65 536 envs, planes once per wave: 48 (76 KB of LDS, two waves per CU) -> 62.5 us in
// steady state and 135 us with every env queued; 32 (51 KB, three per CU) -> 63.9 / 132; 64 with per-lane planes (158 KB,
// one per CU; round 2 and the first half of round 3) -> 60.9 / 165, and 100 whenever the queue outgrew 16 384 envs.
// (SSG_DYN_NONUNI_TU: this file compiled a second time for the per-lane-planes variant alone, with 32 envs per wave — see the end)
#ifdef SSG_DYN_NONUNI_TU
#ifndef SSG_DYN_NONUNI_GRP
#define SSG_DYN_NONUNI_GRP 32
#endif
constexpr int kGrp = SSG_DYN_NONUNI_GRP;
#else
constexpr int kGrp = kDynGrp;
#endif
static_assert(kGrp == 8 || kGrp == 16 || kGrp == 32 || kGrp == 48 || kGrp == 64, "lds[field * kGrp + lane]: at 32 / 64 a lane keeps its LDS banks whatever the field; 48 pays an occasional 2-way conflict on the solver's per-lane body slots");
constexpr int kIter = 10;          // cpSpace iterations
constexpr int kPersist = 3;        // collisionPersistence
constexpr int kMaxGjk = 30, kMaxEpa = 30;
constexpr int kLdsArb = 2;         // arbiter records per env held in LDS; further ones (rare: 98.8 % of the queued envs have <= 2) go to scratch
constexpr int kMaxActive = 8;      // arbiters on one env's solver list
enum { ST_NONE = 0, ST_FIRST = 1, ST_NORMAL = 2, ST_IGNORE = 3, ST_CACHED = 4 };

// ---- per-lane LDS columns ----------------------------------------------------------------------------------
// body slot s (goals 0..ng-1, ships ng..ng+2, the static body ng+3): 8 doubles
enum { B_PX = 0, B_PY, B_VX, B_VY, B_W, B_VBX, B_VBY, B_WB, B_STRIDE };
// ship k extras after the body slots: angle, cos, sin
enum { X_A = 0, X_CA, X_SA, X_STRIDE };
// arbiter record: header + 2 contacts
enum { A_NX = 0, A_NY, A_U, A_INTS /* pid | a << 8 | b << 16 | count << 24 | state << 28 */, A_HASH /* 2 x u32 */, A_CON0 };
enum { AC_R1X = 0, AC_R1Y, AC_R2X, AC_R2Y, AC_NMASS, AC_TMASS, AC_BIAS, AC_JBIAS, AC_JN, AC_JT, AC_STRIDE };
constexpr int A_STRIDE = A_CON0 + 2 * AC_STRIDE;

struct V2 { double x, y; };
__device__ __forceinline__ V2 mk(double x, double y) { V2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ V2 operator+(V2 a, V2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ V2 operator-(V2 a, V2 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ V2 operator*(V2 a, double s) { return mk(a.x * s, a.y * s); }
__device__ __forceinline__ V2 neg(V2 a) { return mk(-a.x, -a.y); }
__device__ __forceinline__ double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ V2 perp(V2 a) { return mk(-a.y, a.x); }
__device__ __forceinline__ V2 rperp(V2 a) { return mk(a.y, -a.x); }
__device__ __forceinline__ double lensq(V2 a) { return dot(a, a); }
__device__ __forceinline__ double len(V2 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ V2 lerp(V2 a, V2 b, double t) { return a * (1.0 - t) + b * t; }
__device__ __forceinline__ V2 normalize(V2 a) { return a * (1.0 / (len(a) + DBL_MIN)); }
__device__ __forceinline__ V2 rotate(V2 a, V2 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// cpfmin / cpfmax / cpfclamp are ternaries in chipmunk_types.h
__device__ __forceinline__ double cmin(double a, double b) { return (a < b) ? a : b; }
__device__ __forceinline__ double cmax(double a, double b) { return (a > b) ? a : b; }
__device__ __forceinline__ double cclamp(double f, double lo, double hi) { return cmin(cmax(f, lo), hi); }
__device__ __forceinline__ double cclamp01(double f) { return cmax(0.0, cmin(f, 1.0)); }

struct BB { double l, b, r, t; };
__device__ __forceinline__ bool bb_hit(const BB &a, const BB &b)
{
    return (a.l <= b.r) & (b.l <= a.r) & (a.b <= b.t) & (b.b <= a.t);
}
__device__ __forceinline__ V2 bb_center(const BB &s) { return lerp(mk(s.l, s.b), mk(s.r, s.t), 0.5); }

struct Sup { V2 p; int i; };
struct Edge { V2 ap, bp; unsigned ah, bh; V2 n; };
// Contact hashes only ever get compared for equality (cpArbiterUpdate): edge point = slot*16 + vertex + 1,
// contact = (hash1 << 8) | hash2, collision free and never 0.
__device__ __forceinline__ unsigned edge_hash(unsigned hashid, int i) { return hashid * 16u + (unsigned)i + 1u; }

// ---- shapes (cpPolyShapeCacheData / cpCircleShapeCacheData evaluated on the fly) ----------------------------
// A ship hull: 5 local vertices then 5 local normals at lds[hoff ..] (wave-uniform constants staged once per
// workgroup: broadcast LDS reads), pose (p, ca, sa).
constexpr int kHullDoubles = 4 * SSG_SHIP_VERTS;
struct ShipShape {
    int hoff;
    V2 p;
    double ca, sa;
    unsigned hashid;
    V2 wv[SSG_SHIP_VERTS]; // world vertices, computed once per shape (cache()): GJK / EPA ask for them a dozen times
    static constexpr bool is_circle = false;
    __device__ __forceinline__ void cache()
    {
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const double hx = lds[hoff + 2 * i], hy = lds[hoff + 2 * i + 1];
            wv[i] = mk(ca * hx + (-sa) * hy + p.x, sa * hx + ca * hy + p.y);
        }
    }
    __device__ __forceinline__ V2 vert(int i) const { return wv[i]; } // i is a compile-time constant at every use
    __device__ __forceinline__ V2 normal(int i) const
    {
        const double nx = lds[hoff + 2 * SSG_SHIP_VERTS + 2 * i], ny = lds[hoff + 2 * SSG_SHIP_VERTS + 2 * i + 1];
        return mk(ca * nx + (-sa) * ny, sa * nx + ca * ny);
    }
    __device__ __forceinline__ BB bb() const
    {
        BB o; o.l = INFINITY; o.r = -INFINITY; o.b = INFINITY; o.t = -INFINITY;
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const V2 v = vert(i);
            o.l = fmin(o.l, v.x); o.r = fmax(o.r, v.x); o.b = fmin(o.b, v.y); o.t = fmax(o.t, v.y);
        }
        return o;
    }
    __device__ __forceinline__ Sup support(V2 n) const // PolySupportPointIndex: first maximum
    {
        double mx = -INFINITY;
        Sup s; s.p = mk(0, 0); s.i = 0;
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const V2 v = vert(i);
            const double d = dot(v, n);
            if (d > mx) { mx = d; s.p = v; s.i = i; }
        }
        return s;
    }
    __device__ __forceinline__ Edge support_edge(V2 n) const // SupportEdgeForPoly
    {
        const int i1 = support(n).i;
        const int i0 = (i1 == 0) ? SSG_SHIP_VERTS - 1 : i1 - 1;
        const int i2 = (i1 == SSG_SHIP_VERTS - 1) ? 0 : i1 + 1;
        V2 v0 = mk(0, 0), v1 = v0, v2 = v0, n1 = v0, n2 = v0;
#pragma unroll
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) { // select without dynamic indexing of the argument arrays
            const V2 v = vert(i), nn = normal(i);
            if (i == i0) v0 = v;
            if (i == i1) { v1 = v; n1 = nn; }
            if (i == i2) { v2 = v; n2 = nn; }
        }
        Edge e;
        if (dot(n, n1) > dot(n, n2)) { e.ap = v0; e.ah = edge_hash(hashid, i0); e.bp = v1; e.bh = edge_hash(hashid, i1); e.n = n1; }
        else { e.ap = v1; e.ah = edge_hash(hashid, i1); e.bp = v2; e.bh = edge_hash(hashid, i2); e.n = n2; }
        return e;
    }
};

// A river bank: static body at the origin.  Its planes (v0, n per vertex) are staged from the map record into this
// lane's LDS columns right before the narrowphase that needs them (all 48 loads in flight at once): GJK / EPA call
// support() a dozen times in a dependent chain, and each call straight from L2 was a round trip.
constexpr int kBankDoubles = 4 * SSG_MAX_HULL;
// UNI: every lane of the wave sits on the same bank record (the queue's map-aligned order, banks of <= 64 records): the
// planes are staged ONCE per wave (field stride 1, broadcast reads) instead of into per-lane columns (field stride kGrp).
template <bool UNI>
struct BankShape {
    static constexpr int kS = UNI ? 1 : kGrp;
    int base; // index in lds[] of (this lane's) staged plane 0
    int n;
    BB box;
    unsigned hashid;
    V2 wv[SSG_MAX_HULL]; // the vertices, fetched from the LDS columns once per shape (cache()): GJK / EPA ask for the support
                         // point a dozen times in a dependent chain, and each call re-read all twelve
    static constexpr bool is_circle = false;
    __device__ __forceinline__ void cache()
    {
#pragma unroll
        for (int i = 0; i < SSG_MAX_HULL; ++i) wv[i] = mk(lds[base + (4 * i) * kS], lds[base + (4 * i + 1) * kS]);
    }
    __device__ __forceinline__ V2 vert(int i) const { return mk(lds[base + (4 * i) * kS], lds[base + (4 * i + 1) * kS]); }
    __device__ __forceinline__ V2 normal(int i) const { return mk(lds[base + (4 * i + 2) * kS], lds[base + (4 * i + 3) * kS]); }
    __device__ __forceinline__ BB bb() const { return box; }
    __device__ __forceinline__ Sup support(V2 nn) const
    {
        double mx = -INFINITY;
        Sup s; s.p = mk(0, 0); s.i = 0;
        // the comparisons run in vertex order over the first n, exactly as PolySupportPointIndex does
#pragma unroll
        for (int i = 0; i < SSG_MAX_HULL; ++i) {
            const double d = dot(wv[i], nn);
            const bool take = (i < n) & (d > mx);
            mx = take ? d : mx; s.p.x = take ? wv[i].x : s.p.x; s.p.y = take ? wv[i].y : s.p.y; s.i = take ? i : s.i;
        }
        return s;
    }
    __device__ __forceinline__ Edge support_edge(V2 nn) const
    {
        const int i1 = support(nn).i;
        const int i0 = (i1 == 0) ? n - 1 : i1 - 1;
        const int i2 = (i1 + 1 == n) ? 0 : i1 + 1;
        Edge e;
        if (dot(nn, normal(i1)) > dot(nn, normal(i2))) {
            e.ap = vert(i0); e.ah = edge_hash(hashid, i0); e.bp = vert(i1); e.bh = edge_hash(hashid, i1); e.n = normal(i1);
        } else {
            e.ap = vert(i1); e.ah = edge_hash(hashid, i1); e.bp = vert(i2); e.bh = edge_hash(hashid, i2); e.n = normal(i2);
        }
        return e;
    }
};

struct CircleShape {
    V2 c;
    double rad;
    static constexpr bool is_circle = true;
    __device__ __forceinline__ BB bb() const { BB o; o.l = c.x - rad; o.b = c.y - rad; o.r = c.x + rad; o.t = c.y + rad; return o; }
    __device__ __forceinline__ Sup support(V2) const { Sup s; s.p = c; s.i = 0; return s; }
};

// ---- cpCollision.c ---------------------------------------------------------------------------------------------
struct Mink { V2 a, b, ab; };
struct Closest { V2 a, b, n; double d; };
struct Info { int count; V2 n; V2 p1[2], p2[2]; unsigned hash[2]; };

template <class SA, class SB>
__device__ __forceinline__ Mink support(const SA &s1, const SB &s2, V2 n)
{
    Mink m;
    m.a = s1.support(neg(n)).p;
    m.b = s2.support(n).p;
    m.ab = m.b - m.a;
    return m;
}
__device__ __forceinline__ double closest_t(V2 a, V2 b)
{
    const V2 delta = b - a;
    return -cclamp(dot(delta, a + b) / lensq(delta), -1.0, 1.0);
}
__device__ __forceinline__ V2 lerp_t(V2 a, V2 b, double t)
{
    const double ht = 0.5 * t;
    return a * (0.5 - ht) + b * (0.5 + ht);
}
__device__ __forceinline__ double closest_dist(V2 v0, V2 v1) { return lensq(lerp_t(v0, v1, closest_t(v0, v1))); }

__device__ __forceinline__ Closest closest_new(const Mink &v0, const Mink &v1)
{
    const double t = closest_t(v0.ab, v1.ab);
    const V2 p = lerp_t(v0.ab, v1.ab, t);
    Closest r;
    r.a = lerp_t(v0.a, v1.a, t);
    r.b = lerp_t(v0.b, v1.b, t);
    const V2 delta = v1.ab - v0.ab;
    const V2 n = normalize(rperp(delta));
    const double d = dot(n, p);
    if (d <= 0.0 || (-1.0 < t && t < 1.0)) {
        r.n = n; r.d = d;
    } else {
        const double d2 = len(p);
        r.n = p * (1.0 / (d2 + DBL_MIN));
        r.d = d2;
    }
    return r;
}

// EPA's growing hull: two buffers of kEpaLds entries {a, b} per lane in LDS columns (ab = b - a is recomputed: the same
// expression that produced it), entries beyond that (practically never) in scratch.
constexpr int kEpaLds = 7;
constexpr int kEpaDoubles = 2 * kEpaLds * 4;
// Development aid (-DSSG_DYN_PROFILE variant builds, tools/c4_stamps.py): cycles of the collide phase by what the wave was
// doing when each interval ENDED, accumulated per wave (the ticks themselves cost ~100 cycles each).
#ifdef SSG_DYN_PROFILE
#define SSG_TICK(mem, cat) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (mem).prof[cat] += n_ - *(mem).last; *(mem).last = n_; } while (0)
#else
#define SSG_TICK(mem, cat) do { } while (0)
#endif
struct EpaMem {
    int base; // index in lds[] of this lane's buffer 0 entry 0 field 0 (field stride 64)
    Mink *ov; // [2][kMaxEpa + 4 - kEpaLds]
    int *cnt; // development counters: [0] gjk iterations [1] epa iterations [2] queries
    unsigned long long *prof, *last; // SSG_DYN_PROFILE: [0] other [1] gjk [2] epa [3] closest/edges/clip [4] push [5] bank staging
    __device__ __forceinline__ Mink get(int buf, int i) const
    {
        if (i < kEpaLds) {
            const int o = base + ((buf * kEpaLds + i) * 4) * kGrp;
            Mink m;
            m.a = mk(lds[o], lds[o + kGrp]); m.b = mk(lds[o + 2 * kGrp], lds[o + 3 * kGrp]);
            m.ab = m.b - m.a;
            return m;
        }
        return ov[buf * (kMaxEpa + 4 - kEpaLds) + i - kEpaLds];
    }
    // ... when every lane's hull is known to fit the LDS entries (wave-uniformly): no scratch alternative behind the access
    __device__ __forceinline__ Mink get_lds(int buf, int i) const
    {
        const int o = base + ((buf * kEpaLds + i) * 4) * kGrp;
        Mink m;
        m.a = mk(lds[o], lds[o + kGrp]); m.b = mk(lds[o + 2 * kGrp], lds[o + 3 * kGrp]);
        m.ab = m.b - m.a;
        return m;
    }
    __device__ __forceinline__ void set_lds(int buf, int i, const Mink &m) const
    {
        const int o = base + ((buf * kEpaLds + i) * 4) * kGrp;
        lds[o] = m.a.x; lds[o + kGrp] = m.a.y; lds[o + 2 * kGrp] = m.b.x; lds[o + 3 * kGrp] = m.b.y;
    }
    __device__ __forceinline__ void set(int buf, int i, const Mink &m) const
    {
        if (i < kEpaLds) {
            const int o = base + ((buf * kEpaLds + i) * 4) * kGrp;
            lds[o] = m.a.x; lds[o + kGrp] = m.a.y; lds[o + 2 * kGrp] = m.b.x; lds[o + 3 * kGrp] = m.b.y;
        } else {
            ov[buf * (kMaxEpa + 4 - kEpaLds) + i - kEpaLds] = m;
        }
    }
};

template <class SA, class SB>
__device__ __forceinline__ Closest epa(const SA &s1, const SB &s2, const Mink &v0, const Mink &v1, const Mink &v2, const EpaMem &mem)
{
    int cur = 0; // buffer holding the hull; the other one receives the rebuilt hull
    int count = 3;
    mem.set_lds(0, 0, v0); mem.set_lds(0, 1, v1); mem.set_lds(0, 2, v2);
    static_assert(kEpaLds >= 4, "the first hull and its first rebuild sit in LDS");
    Closest result;
    bool done = false;
    // One EPA iteration; LDSONLY (a compile-time tag): every lane's hull — and the one it may grow into — fits the LDS entries,
    // so the accessors carry no scratch alternative (the select between the two doubled the instructions of every access).
    auto iterate = [&](auto ldsonly, int iteration) {
        constexpr bool F = decltype(ldsonly)::value;
        auto get = [&](int buf, int i) -> Mink { if constexpr (F) return mem.get_lds(buf, i); else return mem.get(buf, i); };
        auto set = [&](int buf, int i, const Mink &m) { if constexpr (F) mem.set_lds(buf, i, m); else mem.set(buf, i, m); };
        mem.cnt[1]++;
        int mini = 0;
        double min_dist = INFINITY;
        {
            V2 hi = get(cur, count - 1).ab;
            for (int j = 0, i = count - 1; j < count; i = j, ++j) {
                const V2 hj = get(cur, j).ab;
                const double d = closest_dist(hi, hj);
                if (d < min_dist) { min_dist = d; mini = i; }
                hi = hj;
            }
        }
        const int mini1 = (mini + 1 == count) ? 0 : mini + 1; // (mini + 1) % count
        const Mink e0 = get(cur, mini), e1 = get(cur, mini1);
        const Mink p = support(s1, s2, perp(e1.ab - e0.ab));
        const double area2x = cross(e1.ab - e0.ab, (p.ab - e0.ab) + (p.ab - e1.ab));
        if (area2x > 0.0 && iteration < kMaxEpa) {
            int count2 = 1;
            set(cur ^ 1, 0, p);
            V2 h0 = p.ab; // ab of the last entry written to the new hull
            int index = mini1; // (mini + 1 + i) % count, stepped
            for (int i = 0; i < count; ++i) {
                const int next = (index + 1 == count) ? 0 : index + 1;
                const Mink hm = get(cur, index);
                const V2 h1 = hm.ab;
                const V2 h2 = (i + 1 < count) ? get(cur, next).ab : p.ab;
                if (cross(h2 - h0, h1 - h0) > 0.0) { set(cur ^ 1, count2++, hm); h0 = h1; }
                index = next;
            }
            cur ^= 1;
            count = count2;
        } else {
            SSG_TICK(mem, 2);
            result = closest_new(e0, e1);
            done = true;
        }
    };
    for (int iteration = 1;; ++iteration) {
        // (wave-uniform: a hull of `count` entries is rebuilt into at most count + 1)
        if (!__any(count + 1 > kEpaLds)) iterate(std::true_type{}, iteration);
        else iterate(std::false_type{}, iteration);
        if (done) break;
    }
    return result;
}

template <class SA, class SB>
__device__ __forceinline__ Closest gjk(const SA &s1, const SB &s2, const EpaMem &mem)
{
    SSG_TICK(mem, 0);
    const V2 axis = perp(bb_center(s1.bb()) - bb_center(s2.bb())); // cold start (no cached collision id)
    Mink v0 = support(s1, s2, axis);
    Mink v1 = support(s1, s2, neg(axis));
    int iteration = 1;
    mem.cnt[2]++;
    for (;;) {
        mem.cnt[0]++;
        if (iteration > kMaxGjk) { SSG_TICK(mem, 1); return closest_new(v0, v1); }
        const V2 delta = v1.ab - v0.ab;
        if (cross(delta, v0.ab + v1.ab) > 0.0) {
            const Mink tmp = v0; v0 = v1; v1 = tmp; // origin behind the axis: flip, same iteration
            continue;
        }
        const double t = closest_t(v0.ab, v1.ab);
        const V2 n = (-1.0 < t && t < 1.0) ? perp(delta) : neg(lerp_t(v0.ab, v1.ab, t));
        const Mink p = support(s1, s2, n);
        if (cross(v1.ab - p.ab, v1.ab + p.ab) > 0.0 && cross(v0.ab - p.ab, v0.ab + p.ab) < 0.0) {
            SSG_TICK(mem, 1);
            return epa(s1, s2, v0, p, v1, mem);
        }
        if (dot(p.ab, n) <= cmax(dot(v0.ab, n), dot(v1.ab, n))) { SSG_TICK(mem, 1); return closest_new(v0, v1); }
        if (closest_dist(v0.ab, p.ab) < closest_dist(p.ab, v1.ab)) v1 = p; else v0 = p;
        ++iteration;
    }
}

__device__ __forceinline__ void contact_points(const Edge &e1, const Edge &e2, const Closest &points, Info &info)
{
    const double mindist = 0.0 + 0.0;
    if (points.d <= mindist) {
        const V2 n = info.n = points.n;
        const double d_e1_a = cross(e1.ap, n), d_e1_b = cross(e1.bp, n);
        const double d_e2_a = cross(e2.ap, n), d_e2_b = cross(e2.bp, n);
        const double e1_denom = 1.0 / (d_e1_b - d_e1_a + DBL_MIN);
        const double e2_denom = 1.0 / (d_e2_b - d_e2_a + DBL_MIN);
        {
            const V2 p1 = n * 0.0 + lerp(e1.ap, e1.bp, cclamp01((d_e2_b - d_e1_a) * e1_denom));
            const V2 p2 = n * -0.0 + lerp(e2.ap, e2.bp, cclamp01((d_e1_a - d_e2_a) * e2_denom));
            const double dist = dot(p2 - p1, n);
            if (dist <= 0.0) {
                info.p1[info.count] = p1; info.p2[info.count] = p2;
                info.hash[info.count] = e1.ah << 8 | e2.bh;
                info.count++;
            }
        }
        {
            const V2 p1 = n * 0.0 + lerp(e1.ap, e1.bp, cclamp01((d_e2_a - d_e1_a) * e1_denom));
            const V2 p2 = n * -0.0 + lerp(e2.ap, e2.bp, cclamp01((d_e1_b - d_e2_a) * e2_denom));
            const double dist = dot(p2 - p1, n);
            if (dist <= 0.0) {
                info.p1[info.count] = p1; info.p2[info.count] = p2;
                info.hash[info.count] = e1.bh << 8 | e2.ah;
                info.count++;
            }
        }
    }
}

template <class SA, class SB>
__device__ __forceinline__ void collide(const SA &a, const SB &b, Info &info, const EpaMem &mem)
{
    info.count = 0;
    info.n = mk(0, 0);
    if constexpr (SA::is_circle && SB::is_circle) { // CircleToCircle
        const double mindist = a.rad + b.rad;
        const V2 delta = b.c - a.c;
        const double distsq = lensq(delta);
        if (distsq < mindist * mindist) {
            const double dist = sqrt(distsq);
            const V2 n = info.n = (dist != 0.0) ? delta * (1.0 / dist) : mk(1.0, 0.0);
            info.p1[0] = a.c + n * a.rad;
            info.p2[0] = b.c + n * -b.rad;
            info.hash[0] = 0u;
            info.count = 1;
        }
    } else if constexpr (SA::is_circle) { // CircleToPoly
        const Closest points = gjk(a, b, mem);
        const double mindist = a.rad + 0.0;
        if (points.d <= mindist) {
            const V2 n = info.n = points.n;
            info.p1[0] = points.a + n * a.rad;
            info.p2[0] = points.b + n * -0.0;
            info.hash[0] = 0u;
            info.count = 1;
        }
    } else { // PolyToPoly
        const Closest points = gjk(a, b, mem);
        if (points.d - 0.0 - 0.0 <= 0.0) contact_points(a.support_edge(points.n), b.support_edge(neg(points.n)), points, info);
    }
}

// ---- arbiter pair ids (bits of the live mask, rows of the arbiter columns) ----
__device__ __forceinline__ int pid_tb(int k, int s) { return 2 * k + s; }                       // [0, 6)
__device__ __forceinline__ int pid_tt(int j, int k) { return 6 + j + k - 1; }                   // j < k: [6, 9)
__device__ __forceinline__ int pid_gb(int g, int s) { return 9 + 2 * g + s; }                   // [9, 21)
__device__ __forceinline__ int pid_gt(int g, int k) { return 21 + 3 * g + k; }                  // [21, 39)
__device__ __forceinline__ int pid_gg(int h, int g) { return 39 + g * (g - 1) / 2 + h; }        // h < g: [39, 54)

struct DynCols {
    double *f64;
    uint32_t *u32;
    unsigned long long *live;
    uint8_t *flag;
    size_t np;
};

// (re)create the non-player bodies of one env: a fresh pm.Space() after ShipGame.reset + add_default_traffic
__device__ __attribute__((unused)) void dyn_init(const DevCfg &c, const DynCfg &d, const DynCols &col, int e, const double *rec)
{
    double *row = c.dyn_row + (size_t)e * kDynRow; // the row-major shadow the full step loads from
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        double *t = col.f64 + (size_t)(DC_TRAFFIC + 9 * k) * col.np + e;
        t[0 * col.np] = d.tx[k]; t[1 * col.np] = d.ty[k];
        for (int f = 2; f < 9; ++f) t[(size_t)f * col.np] = 0.0;
        row[kDynRowTraffic + 9 * k] = d.tx[k]; row[kDynRowTraffic + 9 * k + 1] = d.ty[k];
        for (int f = 2; f < 9; ++f) row[kDynRowTraffic + 9 * k + f] = 0.0;
        col.f64[(size_t)(DC_TROT + 2 * k) * col.np + e] = 1.0; // cpvforangle(0): the rotation of the angle column (step kernel's collide_ship)
        col.f64[(size_t)(DC_TROT + 2 * k + 1) * col.np + e] = 0.0;
    }
    // all goal centres first, then the stores: a load issued after a store it might alias waits for nothing, but the
    // compiler keeps program order, and one L2 round trip per goal coordinate made this the slowest part of pass 1
    double gxy[2 * SSG_MAX_GOALS];
#pragma unroll
    for (int i = 0; i < 2 * SSG_MAX_GOALS; ++i) gxy[i] = rec[SSG_MAP_OFF_GOALS + i];
#pragma unroll
    for (int g = 0; g < SSG_MAX_GOALS; ++g) {
        if (g >= c.n_goals) break;
        double *q = col.f64 + (size_t)(DC_GOALS + DC_GOAL_COLS * g) * col.np + e;
        q[0 * col.np] = gxy[2 * g];
        q[1 * col.np] = gxy[2 * g + 1];
        for (int f = 2; f < DC_GOAL_COLS; ++f) q[(size_t)f * col.np] = 0.0;
        row[DC_GOAL_COLS * g] = gxy[2 * g]; row[DC_GOAL_COLS * g + 1] = gxy[2 * g + 1];
        for (int f = 2; f < DC_GOAL_COLS; ++f) row[DC_GOAL_COLS * g + f] = 0.0;
    }
    col.live[e] = 0ull;
}

// An arbiter record: LDS columns of this lane (stride 64 doubles) for the first kLdsArb records of an env, scratch
// beyond (rare; never touched otherwise).  No generic pointers: FLAT accesses to LDS stall on both counters.
struct ArbRef {
    int base;   // index of field 0 in lds[] for this lane, or -1
    double *ov; // scratch record otherwise
    __device__ __forceinline__ double get(int f) const { return base >= 0 ? lds[base + f * kGrp] : ov[f]; }
    __device__ __forceinline__ void set(int f, double v) const { if (base >= 0) lds[base + f * kGrp] = v; else ov[f] = v; }
    __device__ __forceinline__ double cget(int k, int f) const { return get(A_CON0 + k * AC_STRIDE + f); }
    __device__ __forceinline__ void cset(int k, int f, double v) const { set(A_CON0 + k * AC_STRIDE + f, v); }
};
// ... the same record when it is known to be one of the first kLdsArb of its env (wave-uniformly: 98.8 % of the queued envs
// have at most two arbiters): no scratch alternative behind every access (the select between the two doubled the instructions
// of cpArbiterUpdate, PreStep and the solver's loads / stores).
struct ArbLds {
    int base;
    __device__ __forceinline__ double get(int f) const { return lds[base + f * kGrp]; }
    __device__ __forceinline__ void set(int f, double v) const { lds[base + f * kGrp] = v; }
    __device__ __forceinline__ double cget(int k, int f) const { return get(A_CON0 + k * AC_STRIDE + f); }
    __device__ __forceinline__ void cset(int k, int f, double v) const { set(A_CON0 + k * AC_STRIDE + f, v); }
};

} // namespace

#ifndef SSG_DYN_NONUNI_TU
// append: the queue of the next full step is valid (the step kernel produced it) and stays so — the envs reset here join it
// (masked ssg_reset between two steps: the RLlib flow resets its done envs this way after every step, ship_env.py:171-184 per env).
// (One launch does the whole reset of a config-4 env: the player's columns and observation rows — reset_env, what reset_kernel runs
// for the 1-ship configs — and the env's other bodies: a masked reset is a launch between every two steps of the RLlib flow.)
__global__ void dyn_reset_kernel(const DevCfg c, const DynCfg d, const uint8_t *__restrict__ mask, const int32_t *__restrict__ map_ids,
                                 double *__restrict__ obs, const int append)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    if (mask && !mask[e]) return;
    DynCols col{c.dyn_f64, c.dyn_u32, c.dyn_live, c.dyn_flag, (size_t)c.n_pad};
    const int m = reset_env(c, e, map_ids, obs);
    const double *rec = c.bank + (size_t)m * SSG_MAP_STRIDE;
    const unsigned old_flag = col.flag[e];
    const int old_map = c.dyn_qmap[e];
    dyn_init(c, d, col, e, rec);
    col.f64[(size_t)(DC_PREV_GOAL + 0) * col.np + e] = rec[SSG_MAP_OFF_SPAWN_GOAL]; // the reset frame's goal
    col.f64[(size_t)(DC_PREV_GOAL + 1) * col.np + e] = rec[SSG_MAP_OFF_SPAWN_GOAL + 1];
    if (!append) { col.flag[e] = 0; return; }
    // A fresh space is stepped in full.  If the env already has an entry in the queue (flag bit 3, set by whoever queued it) under
    // the record it is reset onto, that entry serves (the age only picks the sort bucket); an entry under ANOTHER record is stale
    // — the full step drops entries whose record is not the env's — and a new one is appended.
    const bool queued_here = ((old_flag & 8u) != 0u) & (old_map == m);
    col.flag[e] = 8u;
    if (!queued_here) {
        const unsigned bucket = dyn_bucket_of(0, m);
        const unsigned arrival = atomicAdd(c.dyn_count + (size_t)c.dyn_par * kDynCountWords + dyn_counter_word(bucket), 1u);
        if (arrival < (unsigned)c.n_pad) c.dyn_bucket[(size_t)bucket * col.np + arrival] = e;
        c.dyn_qmap[e] = m;
    }
}

#endif // !SSG_DYN_NONUNI_TU

// 64-bit mixing for the "did this step change anything" test of the full step (inputs vs outputs, no re-reads).
__device__ __forceinline__ unsigned long long mix(unsigned long long h, unsigned long long v)
{
    h = (h ^ v) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}
__device__ __forceinline__ unsigned long long mixd(unsigned long long h, double v)
{
    return mix(h, (unsigned long long)__double_as_longlong(v));
}

// ---------------------------------------------------------------------------------------------------------
// The rest bit.  cpSpaceStep is a deterministic function of the bodies' cpBody fields, the cached arbiters and
// the static banks (the player never pushes anything: PLAYER assumption).  When a full step wrote back exactly the
// bits it had read -- every body field, every arbiter's state / age / contact hashes / accumulated impulses, the live
// mask -- the space is at a fixed point: the next step is the identity.  (That is how a ship resting against a bank
// ends up: the penetration left beyond the slop shrinks by 99.8 % per step until position + bias*dt rounds to the
// position.)  The full step records that as the rest bit (with the bank generation it holds for); while it stands,
// these bodies are skipped (the player's collide_ship test against the parked traffic runs in the step kernel, every step,
// for every env).  A caller that writes the body columns itself must clear the bit with ssg_dyn_invalidate.
// Everything else is appended to the queue of the full step.  In steady state that is the few steps after each reset in
// which ship 1 is pushed out of the left bank, plus whatever the player's goals or a caller stirred up.
// ---------------------------------------------------------------------------------------------------------
constexpr int kClassifyThreads __attribute__((unused)) = 256; // 256 workgroups at 65 536 envs: one per CU (1024-thread workgroups left 192 of the 256 CUs idle)

// Which goals' cached arbiters leave with the goals the player has reached (deferred space.remove, game.py:252).
__device__ __forceinline__ unsigned long long drop_removed_goal_arbiters(unsigned long long live, unsigned gmask, int ng)
{
    for (int g = 0; g < ng; ++g) {
        if ((gmask >> g) & 1u) continue;
        for (int s = 0; s < 2; ++s) live &= ~(1ull << pid_gb(g, s));
        for (int k = 0; k < SSG_N_TRAFFIC; ++k) live &= ~(1ull << pid_gt(g, k));
        for (int h = 0; h < ng; ++h)
            if (h != g) live &= ~(1ull << (h < g ? pid_gg(h, g) : pid_gg(g, h)));
    }
    return live;
}

#ifndef SSG_DYN_NONUNI_TU
// In steady state the step kernel's body role classifies every env for the NEXT step at the end of each step (it holds the
// player's state in registers: shipsim_kernels.hip, role 3) and this kernel does not run.  It runs when the host touched the
// envs in between — ssg_reset, a new bank, ssg_dyn_invalidate, a fresh handle — and rebuilds the queue from the per-env flags.
__global__ __launch_bounds__(kClassifyThreads) void dyn_classify_kernel(const DevCfg c, const DynCfg d)
{
    const int e = blockIdx.x * kClassifyThreads + threadIdx.x;
    const bool valid = e < c.n_envs;
    bool need_full = false;
    unsigned bucket = 0;
    if (valid) {
        DynCols col{c.dyn_f64, c.dyn_u32, c.dyn_live, c.dyn_flag, (size_t)c.n_pad};
        const size_t np = col.np;
        const int ng = c.n_goals;
        // everything this pass can need is requested at once (one memory round trip instead of dependent ones)
        int map_id = c.i32cols[(size_t)ICOL_MAP * np + e];
        int age = c.i32cols[(size_t)ICOL_STEP * np + e]; // steps since the reset: where the env is in its post-reset transient
        unsigned flag = col.flag[e];
        unsigned gm_raw = c.mask[e];
        unsigned long long live0 = col.live[e];
        unsigned long long hash0 = c.dyn_hash[e];
        asm volatile("" : "+v"(map_id), "+v"(age), "+v"(flag), "+v"(gm_raw), "+v"(live0), "+v"(hash0));
        const unsigned gmask = gm_raw & ((1u << ng) - 1u); // goals still in the space
        // rest bit still valid?  It was established for this bank generation; callers that write the body columns
        // themselves clear it with ssg_dyn_invalidate (include/shipsim.h).  A cached arbiter that left with its goal
        // was part of the fixed point: the bodies it touched are stepped again.  (An env the step kernel auto-reset — bit 1 —
        // gets its bodies rebuilt by the full step.)
        const bool rest = ((flag & 6u) == 4u) && (hash0 == (unsigned long long)d.bank_epoch) &&
                          (drop_removed_goal_arbiters(live0, gmask, ng) == live0);
        col.flag[e] = rest ? 4u : (uint8_t)((flag & 6u) | 8u); // bit 3: the env has an entry in the queue of the coming full step
        need_full = !rest;
        bucket = dyn_bucket_of(age, map_id);
        if (need_full) c.dyn_qmap[e] = map_id;
    }
    if (need_full) { // append to the bucket's array (order inside a bucket is irrelevant: every env is stepped on its own)
        unsigned *cnt = c.dyn_count + (size_t)c.dyn_par * kDynCountWords; // THIS step's counter set (zeroed by the host just before)
        const unsigned arrival = atomicAdd(cnt + dyn_counter_word(bucket), 1u);
        c.dyn_bucket[(size_t)bucket * (size_t)c.n_pad + arrival] = e;
    }
}

#endif // !SSG_DYN_NONUNI_TU

// ---------------------------------------------------------------------------------------------------------
// dyn_step_kernel: the full cpSpaceStep of the queued envs, one lane per env, one wave per workgroup.
// UNI: the lanes of a wave share one bank record (see BankShape); chosen by launch_dyn_step.
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int dyn_lane_doubles(int n_goals, bool uni)
{
    return B_STRIDE * (n_goals + SSG_N_TRAFFIC + 1) + X_STRIDE * SSG_N_TRAFFIC + A_STRIDE * kLdsArb + (uni ? 0 : 2 * kBankDoubles) + kEpaDoubles;
}

// MEMO: look the step of this env's state up in the memo table before walking the narrowphase / solver chain, and store
// it there afterwards (shipsim_internal.h, kMemoEntries; UNI launches only: shared bank records are what makes states repeat).
template <bool UNI, bool MEMO>
__global__ __launch_bounds__(64) void dyn_step_kernel(const DevCfg c, const DynCfg d)
{
    const int lane = threadIdx.x;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(); // (development aid, see stamp())
    // This wave's 48 entries, from the bucket counters alone.  Lane m reads the eight age counters of map bucket m; a map's
    // stretch of the (virtual) queue is its eight buckets in age order, rounded up to a whole wave, and the stretches follow each
    // other in map order: an inclusive scan of the rounded lengths tells every lane where its map starts, the one lane whose
    // stretch holds this wave's first slot names the wave's map, and its eight counts place every lane in an age bucket.
    static_assert(kDynMapBuckets == 64 && kDynAgeBuckets == 8, "one lane per map bucket, eight counters each");
    const unsigned *cnt_set = c.dyn_count + (size_t)c.dyn_par * kDynCountWords;
    unsigned ac[kDynAgeBuckets];
#pragma unroll
    for (int j = 0; j < kDynAgeBuckets; ++j) ac[j] = cnt_set[dyn_counter_word((unsigned)(kDynAgeBuckets * lane + j))];
    if (blockIdx.x == 0) { // the OTHER counter set is the step kernel's to fill for the next step: it starts from zero
        unsigned *other = c.dyn_count + (size_t)(c.dyn_par ^ 1) * kDynCountWords;
        for (int i = lane; i < kDynBuckets; i += 64) other[dyn_counter_word((unsigned)i)] = 0u;
    }
    unsigned n_map = 0;
#pragma unroll
    for (int j = 0; j < kDynAgeBuckets; ++j) n_map += min(ac[j], (unsigned)c.n_pad);
    unsigned long long t_h1 = 0ull, t_h2 = 0ull, t_h3 = 0ull; // (development aid, SSG_DYN_STOP=-1: counters / entry / row have arrived)
    if (d.stop_after == -1) { asm volatile("" : "+v"(n_map)); t_h1 = __builtin_amdgcn_s_memtime() - t_start; } // (clamped: garbage counters must not index past an array)
    const unsigned r_map = (n_map + (unsigned)(kGrp - 1)) / (unsigned)kGrp * (unsigned)kGrp;
    unsigned incl = r_map;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o);
        incl += (lane >= o) ? v : 0u;
    }
    const unsigned start = incl - r_map;
    const unsigned slot0 = (unsigned)blockIdx.x * (unsigned)kGrp; // (a stretch is a multiple of kGrp slots: a wave never straddles two maps)
    const unsigned long long owner = __ballot((start <= slot0) & (slot0 < incl));
    if (owner == 0ull) return; // past the end of the queue (block 0 always finds an owner or an empty queue: see below)
    const int qmap = __ffsll((long long)owner) - 1; // wave-uniform: the map bucket of this wave
    const unsigned off = slot0 - (unsigned)__builtin_amdgcn_readlane((int)start, qmap) + (unsigned)lane; // this lane's entry inside the map's stretch
    unsigned base = 0;
    int qage = 0;
    unsigned in_bucket = off;
    bool queued = lane < kGrp;
    {
        unsigned total = 0;
#pragma unroll
        for (int j = 0; j < kDynAgeBuckets; ++j) {
            const unsigned aj = min((unsigned)__builtin_amdgcn_readlane((int)ac[j], qmap), (unsigned)c.n_pad);
            const bool here = (off >= total) & (off < total + aj);
            qage = here ? j : qage;
            base = here ? total : base;
            total += aj;
        }
        queued &= off < total;
        in_bucket = off - base;
    }
    if (!__any(queued)) return; // wave-uniform (only the rounding of a map's last wave)
    // The wave's constants — the four hulls and, UNI, the two banks' planes of the wave's record (known from the counters alone) —
    // are requested WITH the queue entry and parked in registers: staged where the env's row is awaited (hull constants, then a
    // two-trip loop over the planes, each waiting for its own loads before its LDS write), they were three dependent round trips
    // in front of every wave's first instruction of real work.
    double hc[1 + SSG_N_TRAFFIC] = {0.0, 0.0, 0.0, 0.0}, plq[2] = {0.0, 0.0};
    double bk[2][5]; // both banks' vertex counts and AABBs
    if (lane < kHullDoubles) {
        const int i = lane % (2 * SSG_SHIP_VERTS);
        const bool nr = lane >= 2 * SSG_SHIP_VERTS;
        hc[0] = nr ? c.nrm[i] : c.hull[i];
#pragma unroll
        for (int k = 0; k < SSG_N_TRAFFIC; ++k) hc[1 + k] = nr ? d.tnrm[k][i] : d.thull[k][i];
    }
    if (UNI) {
        const double *rec_u = c.bank + (size_t)qmap * SSG_MAP_STRIDE;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = lane + 64 * t;
            if (q < 2 * kBankDoubles) {
                const int sd = q / kBankDoubles, j = (q % kBankDoubles) / 4, f = q % 4;
                plq[t] = rec_u[SSG_MAP_OFF_PLANES + sd * SSG_MAX_HULL * SSG_PLANE_DOUBLES + SSG_PLANE_DOUBLES * j + f];
            }
        }
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            bk[sd][0] = rec_u[SSG_MAP_OFF_COUNTS + sd];
#pragma unroll
            for (int f = 0; f < 4; ++f) bk[sd][1 + f] = rec_u[SSG_MAP_OFF_AABB + 4 * sd + f];
        }
    }
    const int e_q = queued ? c.dyn_bucket[(size_t)(kDynAgeBuckets * qmap + qage) * (size_t)c.n_pad + in_bucket] : 0;
    asm volatile("" : "+v"(hc[0]), "+v"(hc[1]), "+v"(hc[2]), "+v"(hc[3]), "+v"(plq[0]), "+v"(plq[1]));
    if (UNI) {
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int f = 0; f < 5; ++f) asm volatile("" : "+v"(bk[sd][f]));
    }
    if (d.stop_after == -1) { int eq_ = e_q; asm volatile("" : "+v"(eq_)); t_h2 = __builtin_amdgcn_s_memtime() - t_start; }
    queued &= (e_q >= 0) & (e_q < c.n_envs);
    const int e = queued ? e_q : 0;
    const int lane_doubles = dyn_lane_doubles(c.n_goals, UNI);
    const int cbase = kGrp * lane_doubles;
    const int sbank = cbase + kHullDoubles * (1 + SSG_N_TRAFFIC); // UNI: the wave's two banks, plane-major as in the per-lane columns
    DynCols col{c.dyn_f64, c.dyn_u32, c.dyn_live, c.dyn_flag, (size_t)c.n_pad};
    const size_t np = col.np;
    // Everything the step reads from memory about this env — goal mask, flags, live-arbiter mask, the body row, (UNI) the wave's bank planes — is requested NOW, in one batch behind the queue entry.  (Requested where
    // they were used these were four dependent round trips at the head of every wave's chain: the entry, the env's map id,
    // the planes of that record, then the body state.)
    int map_col = c.i32cols[(size_t)ICOL_MAP * np + e];
    unsigned gm_raw = c.mask[e];
    unsigned flag_raw = col.flag[e];
    unsigned long long live_raw = col.live[e];
    double rw[kDynRow];
    {
        // from the env's row of the row-major shadow: 40 16-byte loads over five cache lines of this lane, instead of 75 column
        // gathers over 75 x 64 lines per wave (the bucketed queue scatters a wave's envs over the whole batch)
        const double2 *row2 = reinterpret_cast<const double2 *>(c.dyn_row + (size_t)e * kDynRow);
#pragma unroll
        for (int i = 0; i < kDynRow / 2; ++i) { const double2 rv = row2[i]; rw[2 * i] = rv.x; rw[2 * i + 1] = rv.y; }
    }
    if (lane < kHullDoubles) { // the hull constants ([0] the player, [1..3] the traffic ships; vertices then normals) at lds[cbase ..]
#pragma unroll
        for (int k = 0; k < 1 + SSG_N_TRAFFIC; ++k) lds[cbase + kHullDoubles * k + lane] = hc[k];
    }
    int map_id = map_col;
    if (UNI) {
        if (!__any(queued)) return;
        map_id = qmap; // (banks of at most 64 records: the wave's map bucket IS its record — known before the entries are)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (lane + 64 * t < 2 * kBankDoubles) lds[sbank + lane + 64 * t] = plq[t];
    }
    asm volatile("" : "+v"(map_col), "+v"(gm_raw), "+v"(flag_raw), "+v"(live_raw));
#pragma unroll
    for (int i = 0; i < kDynRow; ++i) asm volatile("" : "+v"(rw[i]));
    if (!UNI) map_id = map_col;
    if (d.stop_after == -1) t_h3 = __builtin_amdgcn_s_memtime() - t_start;
    // an entry queued under another record than the env now sits on is stale: a masked ssg_reset moved the env after it was
    // queued (and queued it again under its new record); entries under the env's record are all equivalent
    if (UNI) queued &= (map_col & (kDynMapBuckets - 1)) == qmap;
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0); one wave per workgroup: the LDS writes above are visible to its lanes
    __builtin_amdgcn_wave_barrier();
    if (!queued) return;
    const double dt = c.dt;
    const int ng = c.n_goals;
    const double *rec = c.bank + (size_t)map_id * SSG_MAP_STRIDE;
    // development aid (SSG_DYN_STOP=-1): phase stamps of this lane's wave into the unused arbiter rows of pair 50..53
    auto stamp = [&](int i) {
        if (d.stop_after == -1)
            col.f64[(size_t)(DC_ARB + 4 * 50 + i) * np + e] = (double)(__builtin_amdgcn_s_memtime() - t_start);
    };
    const unsigned gmask = gm_raw & ((1u << ng) - 1u); // goals still in the space
    // bit 1 of the flag: the step kernel auto-reset this env at the end of the last step — a fresh pm.Space(): its bodies are
    // rebuilt here (ShipGame.reset + add_default_traffic), nothing of the old episode is read
    const bool fresh = (flag_raw & 2u) != 0u;
    // deferred space.remove of the goals the player reached last step (game.py:252): their cached arbiters go with them
    unsigned long long live = fresh ? 0ull : drop_removed_goal_arbiters(live_raw, gmask, ng);
    const unsigned long long live0 = live;
    // The records of the cached arbiters (a handful of pairs at most: the bits of `live`) are requested NOW, with the body row:
    // fetched where cpArbiterUpdate needs them (push(), below; the ageing loop after the narrowphase) each was a dependent
    // L2 / HBM round trip in the middle of the collide phase — 8 k cycles of a wave's chain with two arbiters.
    constexpr int kPre = 4;
    static_assert(kPre == kMemoArbIn, "the memo key holds exactly the cached arbiters that are prefetched here");
    static_assert(kMaxActive <= kMemoArbOut, "the memo value has one slot per arbiter the solver's list can hold");
    int ppid[kPre];
    unsigned pmeta[kPre], phh[kPre];
    double pacc[kPre][4];
    {
        unsigned long long lv = live;
#pragma unroll
        for (int i = 0; i < kPre; ++i) {
            ppid[i] = -1; pmeta[i] = 0u; phh[i] = 0u;
#pragma unroll
            for (int f = 0; f < 4; ++f) pacc[i][f] = 0.0;
            if (lv) {
                const int pid = __ffsll((long long)lv) - 1;
                lv &= lv - 1ull;
                ppid[i] = pid;
                pmeta[i] = col.u32[(size_t)(DU_META + pid) * np + e];
                if (pid < kPolyPairs) phh[i] = col.u32[(size_t)(DU_HASH + pid) * np + e];
#pragma unroll
                for (int f = 0; f < 4; ++f) pacc[i][f] = col.f64[(size_t)(DC_ARB + 4 * pid + f) * np + e];
            }
        }
    }
    // (a cached arbiter beyond the first kPre of the mask — never seen — is fetched on the spot)
    auto arb_old = [&](int pid, unsigned &meta, unsigned &hh, double &a0, double &a1, double &a2, double &a3) {
        bool found = false;
#pragma unroll
        for (int i = 0; i < kPre; ++i)
            if (ppid[i] == pid) { meta = pmeta[i]; hh = phh[i]; a0 = pacc[i][0]; a1 = pacc[i][1]; a2 = pacc[i][2]; a3 = pacc[i][3]; found = true; }
        if (!found) {
            meta = col.u32[(size_t)(DU_META + pid) * np + e];
            hh = (pid < kPolyPairs) ? col.u32[(size_t)(DU_HASH + pid) * np + e] : 0u;
            const double *acc = col.f64 + (size_t)(DC_ARB + 4 * pid) * np + e;
            a0 = acc[0 * np]; a1 = acc[1 * np]; a2 = acc[2 * np]; a3 = acc[3 * np];
        }
    };
    if (d.stop_after == -1) { // (development aid: the head — queue entry, row, cached arbiters — has arrived)
        unsigned long long pk_ = 0ull;
#pragma unroll
        for (int i = 0; i < kPre; ++i) pk_ += (unsigned long long)pmeta[i] + (unsigned long long)__double_as_longlong(pacc[i][0]);
        asm volatile("" : "+v"(pk_));
        stamp(15);
        col.f64[(size_t)(DC_ARB + 4 * 45 + 0) * np + e] = (double)t_h1; col.f64[(size_t)(DC_ARB + 4 * 45 + 1) * np + e] = (double)t_h2;
        col.f64[(size_t)(DC_ARB + 4 * 45 + 2) * np + e] = (double)t_h3;
    }
    // Did this step write back anything but the bits it read?  The body columns are compared directly: what was read stays
    // in registers until the write-back (this kernel runs one wave per SIMD, 512 VGPRs to spare; hashing both sides cost
    // ~150 64-bit multiplies per step).  The arbiters' accumulated impulses, touched for a few pairs only, go through
    // 64-bit hashes.
    bool changed = false;
    unsigned long long ain = 0ull, aout = 0ull; // order-independent (xor) over the touched arbiters
    auto arb_hash = [&](int pid, unsigned meta, unsigned hh, double j0, double j1, double t0, double t1) -> unsigned long long {
        return mixd(mixd(mixd(mixd(mix(mix(mix(0x7F4A7C15ull, (unsigned)pid), meta), hh), j0), j1), t0), t1);
    };

    // ---- LDS columns of this lane ---------------------------------------------------------------------------
    const int slot_ship0 = ng, slot_static = ng + SSG_N_TRAFFIC;
    const int xbase = B_STRIDE * (slot_static + 1), abase = xbase + X_STRIDE * SSG_N_TRAFFIC;
    auto L = [&](int f) -> double & { return lds[f * kGrp + lane]; };
    auto BF = [&](int slot, int f) -> double & { return lds[(B_STRIDE * slot + f) * kGrp + lane]; };
    for (int f = 0; f < B_STRIDE; ++f) BF(slot_static, f) = 0.0; // cpBodyNewStatic at the origin
    auto m_inv_of = [&](int slot) -> double { return slot < slot_ship0 ? d.goal_m_inv : (slot < slot_static ? d.t_m_inv : 0.0); };
    auto i_inv_of = [&](int slot) -> double {
        if (slot < slot_ship0) return d.goal_i_inv;
        const int k = slot - slot_ship0;
        return k == 0 ? d.t_i_inv[0] : (k == 1 ? d.t_i_inv[1] : (k == 2 ? d.t_i_inv[2] : 0.0));
    };

    // ---- (1) load + cpBodyUpdatePosition ------------------------------------------------------------------------
    // every body column of this env is requested at once (one memory round trip; per goal and ship it was five dependent ones)
    // (the player's six columns are requested with everything else: its pose after its own cpBodyUpdatePosition)
    double gin[SSG_MAX_GOALS][DC_GOAL_COLS], tin[SSG_N_TRAFFIC][9];
    {
        static_assert(DC_GOAL_COLS * SSG_MAX_GOALS == kDynRowTraffic && kDynRowTraffic + 9 * SSG_N_TRAFFIC <= kDynRow && kDynRow % 2 == 0, "row layout");
#pragma unroll
        for (int g = 0; g < SSG_MAX_GOALS; ++g)
#pragma unroll
            for (int f = 0; f < DC_GOAL_COLS; ++f) gin[g][f] = rw[DC_GOAL_COLS * g + f];
#pragma unroll
        for (int k = 0; k < SSG_N_TRAFFIC; ++k)
#pragma unroll
            for (int f = 0; f < 9; ++f) tin[k][f] = rw[kDynRowTraffic + 9 * k + f];
        if (fresh) { // dyn_init's values, in registers; the write-back below stores every field of such an env
            double gxy[2 * SSG_MAX_GOALS];
#pragma unroll
            for (int i = 0; i < 2 * SSG_MAX_GOALS; ++i) gxy[i] = rec[SSG_MAP_OFF_GOALS + i];
#pragma unroll
            for (int g = 0; g < SSG_MAX_GOALS; ++g) {
                gin[g][0] = gxy[2 * g]; gin[g][1] = gxy[2 * g + 1];
#pragma unroll
                for (int f = 2; f < DC_GOAL_COLS; ++f) gin[g][f] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
                tin[k][0] = d.tx[k]; tin[k][1] = d.ty[k];
#pragma unroll
                for (int f = 2; f < 9; ++f) tin[k][f] = 0.0;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < SSG_MAX_GOALS; ++g) {
        if (g >= ng) continue;
        if (!((gmask >> g) & 1u)) continue;
        const V2 p = mk(gin[g][0], gin[g][1]), v = mk(gin[g][2], gin[g][3]), vb = mk(gin[g][4], gin[g][5]);
        const double w = gin[g][6];
        const V2 pn = p + (v + vb) * dt; // (the angle of a circle body is never read)
        BF(g, B_PX) = pn.x; BF(g, B_PY) = pn.y; BF(g, B_VX) = v.x; BF(g, B_VY) = v.y; BF(g, B_W) = w;
        BF(g, B_VBX) = 0.0; BF(g, B_VBY) = 0.0; BF(g, B_WB) = 0.0;
    }
#pragma unroll
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        const V2 p = mk(tin[k][0], tin[k][1]), v = mk(tin[k][3], tin[k][4]), vb = mk(tin[k][6], tin[k][7]);
        const double a = tin[k][2], w = tin[k][5], wb = tin[k][8];
        const V2 pn = p + (v + vb) * dt;
        const double an = a + (w + wb) * dt;
        double sa, ca;
        sincos_body(an, &sa, &ca);
        const int s = slot_ship0 + k;
        BF(s, B_PX) = pn.x; BF(s, B_PY) = pn.y; BF(s, B_VX) = v.x; BF(s, B_VY) = v.y; BF(s, B_W) = w;
        BF(s, B_VBX) = 0.0; BF(s, B_VBY) = 0.0; BF(s, B_WB) = 0.0;
        L(xbase + X_STRIDE * k + X_A) = an; L(xbase + X_STRIDE * k + X_CA) = ca; L(xbase + X_STRIDE * k + X_SA) = sa;
    }
    auto ship_shape = [&](int k) -> ShipShape {
        ShipShape s;
        s.hoff = cbase + kHullDoubles * (1 + k);
        s.p = mk(BF(slot_ship0 + k, B_PX), BF(slot_ship0 + k, B_PY));
        s.ca = L(xbase + X_STRIDE * k + X_CA); s.sa = L(xbase + X_STRIDE * k + X_SA);
        s.hashid = (unsigned)(2 + SSG_MAX_GOALS + k);
        s.cache();
        return s;
    };
    // both banks' vertex counts and AABBs (UNI: requested with the queue entry, above; per-lane records: one round trip here)
    if (!UNI) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bk[s][0] = rec[SSG_MAP_OFF_COUNTS + s];
#pragma unroll
            for (int f = 0; f < 4; ++f) bk[s][1 + f] = rec[SSG_MAP_OFF_AABB + 4 * s + f];
        }
    }
    const int bbase = (abase + A_STRIDE * kLdsArb) * kGrp + lane; // !UNI: this lane's staged bank planes; EPA's hull follows
    Mink epa_ov[2 * (kMaxEpa + 4 - kEpaLds)];
    EpaMem emem;
    int dbg_cnt[3] = {0, 0, 0};
    unsigned long long prof_acc[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, prof_last = 0ull;
#ifdef SSG_DYN_PROFILE
    unsigned long long type_acc[5] = {0ull, 0ull, 0ull, 0ull, 0ull}, type_t0 = 0ull; // cycles per pair type: gb gg tb gt tt
    double type_cnt[5] = {0.0, 0.0, 0.0, 0.0, 0.0};                                   // ... and the wave's trips of that type
#define SSG_TYPE_BEGIN() do { type_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define SSG_TYPE_END(t) do { type_acc[t] += __builtin_amdgcn_s_memtime() - type_t0; type_cnt[t] += 1.0; } while (0)
#else
#define SSG_TYPE_BEGIN() do { } while (0)
#define SSG_TYPE_END(t) do { } while (0)
#endif
    emem.base = (abase + A_STRIDE * kLdsArb) * kGrp + lane + (UNI ? 0 : 2 * kBankDoubles * kGrp); emem.ov = epa_ov; emem.cnt = dbg_cnt; emem.prof = prof_acc; emem.last = &prof_last;
    auto bank_box = [&](int s) -> BB {
        BB o;
        o.l = s ? bk[1][1] : bk[0][1]; o.b = s ? bk[1][2] : bk[0][2]; o.r = s ? bk[1][3] : bk[0][3]; o.t = s ? bk[1][4] : bk[0][4];
        return o;
    };
    if (!UNI) {
        // Both banks' planes go to this lane's LDS columns up front, requested together with the body row: staged on demand
        // (one bank at a time, again whenever the pair loops switched side) the 48 gathers from the map record were an exposed
        // L2 round trip in front of every narrowphase query.
        // (16-byte loads: a divergent gather costs the address path one request per lane whatever its width; records are
        // 8-byte aligned, gfx950 serves the unaligned dwordx4)
        typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));
        double tmp[2 * kBankDoubles];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < SSG_MAX_HULL; ++j)
#pragma unroll
                for (int f = 0; f < 4; f += 2) { // all 12 slots exist in the record
                    const double2_a8 v = *reinterpret_cast<const double2_a8 *>(rec + SSG_MAP_OFF_PLANES + s * SSG_MAX_HULL * SSG_PLANE_DOUBLES + SSG_PLANE_DOUBLES * j + f);
                    tmp[s * kBankDoubles + 4 * j + f] = v.x; tmp[s * kBankDoubles + 4 * j + f + 1] = v.y;
                }
#pragma unroll
        for (int q = 0; q < 2 * kBankDoubles; ++q) lds[bbase + q * kGrp] = tmp[q];
    }
    auto bank_shape = [&](int s) -> BankShape<UNI> {
        BankShape<UNI> b;
        b.base = UNI ? sbank + s * kBankDoubles : bbase + s * kBankDoubles * kGrp;
        b.n = (int)(s ? bk[1][0] : bk[0][0]);
        b.box = bank_box(s);
        b.hashid = (unsigned)s;
        b.cache();
        return b;
    };
    auto goal_shape = [&](int g) -> CircleShape { CircleShape s; s.c = mk(BF(g, B_PX), BF(g, B_PY)); s.rad = c.goal_r; return s; };

    stamp(0);
    if (d.stop_after == 1) return;
    // (collide_ship, the player against the traffic ships, is the step kernel's: it reads the poses this step writes)
    stamp(1);
    if (d.stop_after == 2) return;
    prof_last = __builtin_amdgcn_s_memtime();
    // ---- (3) collide, canonical order ---------------------------------------------------------------------------
    double ovf[(kMaxActive - kLdsArb) * A_STRIDE]; // records beyond the LDS ones: scratch, touched only when used
    auto arb = [&](int i) -> ArbRef {
        ArbRef r;
        r.base = (i < kLdsArb) ? (abase + A_STRIDE * i) * kGrp + lane : -1;
        r.ov = &ovf[(i < kLdsArb ? 0 : i - kLdsArb) * A_STRIDE];
        return r;
    };
    auto arb_lds = [&](int i) -> ArbLds { ArbLds r; r.base = (abase + A_STRIDE * i) * kGrp + lane; return r; };
    int n_act = 0;
    unsigned long long touched = 0ull;

    // cpSpaceCollideShapes after the narrowphase found contacts: fetch / create the cached arbiter, cpArbiterUpdate
    auto push = [&](const Info &info, int a, int b, int pid, double u) {
        SSG_TICK(emem, 3);
        if (info.count == 0 || n_act >= kMaxActive) return;
        int state = ST_NONE, old_count = 0;
        unsigned old_hash[2] = {0u, 0u};
        double old_jn[2] = {0.0, 0.0}, old_jt[2] = {0.0, 0.0};
        if ((live >> pid) & 1ull) {
            unsigned meta = 0u, hh = 0u;
            arb_old(pid, meta, hh, old_jn[0], old_jn[1], old_jt[0], old_jt[1]);
            state = meta & 7u;
            old_count = (meta >> 5) & 3u;
            if (pid < kPolyPairs) { old_hash[0] = hh & 0xFFFFu; old_hash[1] = hh >> 16; }
            ain ^= arb_hash(pid, meta & ~0x18u, (pid < kPolyPairs) ? (old_hash[0] | old_hash[1] << 16) : 0u, old_jn[0], old_jn[1],
                            old_jt[0], old_jt[1]); // (age bits are 0 for an arbiter touched every step; a cached one changes anyway)
            if ((meta >> 3) & 3u) changed = true;
            if (state == ST_FIRST) state = ST_NORMAL; // it was on last step's solver list
        }
        if (state == ST_NONE) { state = ST_FIRST; old_count = 0; } // cpArbiterInit
        // (wave-uniform: does every lane's new record still sit in LDS?)
        const bool lds_only = !__any(n_act >= kLdsArb);
        const int slot_i = n_act++;
        auto fill = [&](const auto A) {
        A.set(A_NX, info.n.x); A.set(A_NY, info.n.y); A.set(A_U, u);
        const V2 pa = mk(BF(a, B_PX), BF(a, B_PY)), pb = mk(BF(b, B_PX), BF(b, B_PY));
        unsigned hh[2] = {0u, 0u};
        for (int i = 0; i < info.count; ++i) { // cpArbiterUpdate
            const V2 r1 = info.p1[i] - pa, r2 = info.p2[i] - pb;
            double jn = 0.0, jt = 0.0;
            for (int j = 0; j < old_count; ++j)
                if (info.hash[i] == old_hash[j]) { jn = old_jn[j]; jt = old_jt[j]; }
            A.cset(i, AC_R1X, r1.x); A.cset(i, AC_R1Y, r1.y); A.cset(i, AC_R2X, r2.x); A.cset(i, AC_R2Y, r2.y);
            A.cset(i, AC_JN, jn); A.cset(i, AC_JT, jt);
            hh[i] = info.hash[i];
        }
        if (state == ST_CACHED) state = ST_FIRST;
        const unsigned ints = (unsigned)pid | ((unsigned)a << 8) | ((unsigned)b << 16) | ((unsigned)info.count << 24) | ((unsigned)state << 28);
        A.set(A_INTS, __longlong_as_double((long long)ints));
        A.set(A_HASH, __longlong_as_double((long long)(((unsigned long long)hh[1] << 32) | hh[0])));
        };
        if (lds_only) fill(arb_lds(slot_i)); else fill(arb(slot_i));
        touched |= 1ull << pid;
        live |= 1ull << pid;
        SSG_TICK(emem, 4);
    };

    // ---- broadphase: every candidate pair's cpBBIntersects (queryReject) at once, in registers --------------------------
    // cpSpaceStep visits the pairs in the canonical order of the named ORDER assumption: per goal g its two bank pairs, then
    // the goals h < g; then per ship k its two bank pairs, the goals, the ships j < k.  Bit o of `cand` = the o-th pair of that
    // order passed the box test; the narrowphase below takes the set bits in ascending order, so arbiters reach the solver's
    // list in the same order as the rolled pair loops produced them — which paid ~44 dependent LDS round trips and branches
    // per env for the one or two pairs that survive.
    constexpr int kGoalBlock = 2 * SSG_MAX_GOALS + SSG_MAX_GOALS * (SSG_MAX_GOALS - 1) / 2; // 27 bits: goal g starts at 2g + g(g-1)/2
    auto o_goal = [](int g) { return 2 * g + g * (g - 1) / 2; };
    auto o_ship = [=](int k) { return kGoalBlock + (2 + SSG_MAX_GOALS) * k + k * (k - 1) / 2; };
    static_assert(kGoalBlock + 3 * (2 + SSG_MAX_GOALS) + 3 <= 64 && SSG_N_TRAFFIC == 3, "candidate mask layout");
    unsigned long long cand = 0ull;
    {
        BB gbx[SSG_MAX_GOALS], kbx[SSG_N_TRAFFIC];
        const BB bank0 = bank_box(0), bank1 = bank_box(1);
#pragma unroll
        for (int g = 0; g < SSG_MAX_GOALS; ++g) {
            CircleShape cg;
            cg.c = (g < ng) ? mk(BF(g < ng ? g : 0, B_PX), BF(g < ng ? g : 0, B_PY)) : mk(0, 0);
            cg.rad = c.goal_r;
            gbx[g] = cg.bb();
        }
#pragma unroll
        for (int k = 0; k < SSG_N_TRAFFIC; ++k) kbx[k] = ship_shape(k).bb();
#pragma unroll
        for (int g = 0; g < SSG_MAX_GOALS; ++g) {
            const bool pg = (g < ng) & (bool)((gmask >> g) & 1u);
            cand |= (unsigned long long)(pg & bb_hit(gbx[g], bank0)) << (o_goal(g) + 0);
            cand |= (unsigned long long)(pg & bb_hit(gbx[g], bank1)) << (o_goal(g) + 1);
#pragma unroll
            for (int h = 0; h < g; ++h)
                cand |= (unsigned long long)(pg & (bool)((gmask >> h) & 1u) & bb_hit(gbx[h], gbx[g])) << (o_goal(g) + 2 + h);
        }
#pragma unroll
        for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
            cand |= (unsigned long long)bb_hit(kbx[k], bank0) << (o_ship(k) + 0);
            cand |= (unsigned long long)bb_hit(kbx[k], bank1) << (o_ship(k) + 1);
#pragma unroll
            for (int g = 0; g < SSG_MAX_GOALS; ++g)
                cand |= (unsigned long long)((g < ng) & (bool)((gmask >> g) & 1u) & bb_hit(gbx[g], kbx[k])) << (o_ship(k) + 2 + g);
#pragma unroll
            for (int j = 0; j < k; ++j) cand |= (unsigned long long)bb_hit(kbx[j], kbx[k]) << (o_ship(k) + 2 + SSG_MAX_GOALS + j);
        }
    }
    SSG_TICK(emem, 5); // (profile builds: category 5 = the broadphase)
    stamp(10);
    // ---- the memo: has the step of exactly this state been computed before? ------------------------------------------------
    // Key = everything the rest of the step reads: the bank record, the three ships' fields, the fields of the goals that take
    // part (still in the space, and moving or a broadphase candidate this step; the others are stepped by the identity and seen
    // by nobody), the cached arbiters.  A lane whose state is in the table copies the stored result — verified word for word
    // against the stored key, so it writes the bits the computation below would have written — and leaves; the rest compute and
    // store.  (Entries written by this launch are not read by it: `born`.)
    using u64 = unsigned long long;
    auto dbits = [](double v) -> u64 { return (u64)__double_as_longlong(v); };
    bool memo_try = false;    // this lane computes, then stores its result in the table
    u64 *memo_ent = nullptr;  // ... in this entry,
    u64 memo_old = 1ull; // the claiming CAS's answer ^ the expected tag: 0 = claimed
    unsigned memo_incl = 0u;
    u64 memo_hdr = 0ull;
    u64 memo_aged[kMemoAged] = {0ull, 0ull, 0ull, 0ull};
    int memo_n_aged = 0;
    // The key's sections, as (index of the section's first word, words) pairs a visitor is called with — two words at a time,
    // the entry's 16-byte granule.  Ships: always.  Goals: the ones that take part.  Arbiters: the env's cached ones.
    auto key_ship_word = [&](int j) -> u64 { return j < 9 * SSG_N_TRAFFIC ? dbits(tin[(j < 27 ? j : 0) / 9][(j < 27 ? j : 0) % 9]) : 0ull; };
    auto key_arb_word = [&](int a, int f) -> u64 {
        if (f == 0) return (u64)(unsigned)ppid[a] | ((u64)pmeta[a] << 8) | ((u64)phh[a] << 32);
        return f <= 4 ? dbits(pacc[a][f <= 4 ? f - 1 : 0]) : 0ull;
    };
    auto key_visit = [&](auto &&f2 /* (word index, w0, w1) */, const unsigned incl, const int n_live) {
        f2(0, memo_hdr, live0);
#pragma unroll
        for (int j = 0; j < 28; j += 2) f2(kMemoKeyShips + j, key_ship_word(j), key_ship_word(j + 1));
#pragma unroll
        for (int g = 0; g < SSG_MAX_GOALS; ++g) {
            if (!((incl >> g) & 1u)) continue;
#pragma unroll
            for (int f = 0; f < DC_GOAL_COLS; f += 2) f2(kMemoKeyGoals + DC_GOAL_COLS * g + f, dbits(gin[g][f]), dbits(gin[g][f + 1]));
        }
#pragma unroll
        for (int a = 0; a < kMemoArbIn; ++a) {
            if (a >= n_live) continue;
#pragma unroll
            for (int f = 0; f < kMemoArbWords; f += 2) f2(kMemoKeyArbs + kMemoArbWords * a + f, key_arb_word(a, f), key_arb_word(a, f + 1));
        }
    };
    int memo_n_live = 0;
    if constexpr (MEMO) {
        unsigned incl = 0u;
#pragma unroll
        for (int g = 0; g < SSG_MAX_GOALS; ++g) {
            // candidate bits goal g appears in: its two bank pairs, the goals h < g, goal g in the rows of the goals g' > g, the ships
            u64 inv = (3ull << o_goal(g));
#pragma unroll
            for (int h = 0; h < g; ++h) inv |= 1ull << (o_goal(g) + 2 + h);
#pragma unroll
            for (int g2 = g + 1; g2 < SSG_MAX_GOALS; ++g2) inv |= 1ull << (o_goal(g2) + 2 + g);
#pragma unroll
            for (int k = 0; k < SSG_N_TRAFFIC; ++k) inv |= 1ull << (o_ship(k) + 2 + g);
            const bool present = (g < ng) & (bool)((gmask >> g) & 1u);
            bool moving = false;
#pragma unroll
            for (int f = 2; f < DC_GOAL_COLS; ++f) moving |= gin[g][f] != 0.0;
            incl |= (present & (moving | ((cand & inv) != 0ull))) ? (1u << g) : 0u;
        }
        memo_incl = incl;
        const int n_live = __popcll(live0);
        memo_n_live = n_live;
        const bool memo_ok = n_live <= kMemoArbIn; // (the first kMemoArbIn cached arbiters are the ones held in ppid / pacc)
        memo_hdr = (u64)((unsigned)map_id & 0xFFFFu) | ((u64)incl << 16) | ((u64)(unsigned)n_live << 24) | ((u64)(d.memo_fp & 0xFFFFu) << 32) |
                   ((u64)(d.bank_epoch & 0xFFFFu) << 48); // (banks of up to 65 536 records: launch_dyn_step)
        // tag: two rotate-xor lanes over the key, one final mix (the key itself is compared on a hit: the tag only has to spread)
        u64 h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
        key_visit([&](int, u64 w0, u64 w1) {
            h0 = ((h0 << 7) | (h0 >> 57)) ^ w0;
            h1 = ((h1 << 11) | (h1 >> 53)) ^ w1;
        }, incl, memo_ok ? n_live : 0);
        const u64 hh = mix(mix(h0, h1), h0 >> 32);
        const u64 gen = (u64)(c.dyn_memo_gen & 0xFFu);
        const u64 tag = (hh & ~0xFFull) | gen;
        const unsigned slot0 = (unsigned)(hh >> 24);
        // the headers of the probe sequence, one round trip
        u64 ptag[kMemoProbes], prdy[kMemoProbes];
#pragma unroll
        for (int p = 0; p < kMemoProbes; ++p) {
            const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(c.dyn_memo + (size_t)((slot0 + (unsigned)p) & (unsigned)(kMemoEntries - 1)) * kMemoStride);
            ptag[p] = a.x; prdy[p] = a.y;
        }
        int cand_p = -1, free_p = -1;
        bool claimed = false; // somebody is writing (or has written, this launch) an entry with this tag
        u64 old_tag = 0ull;
#pragma unroll
        for (int p = kMemoProbes - 1; p >= 0; --p) {
            const bool mine = ptag[p] == tag;
            // `ready` = (launch that completed the entry) << 8 | generation, stored last: usable = completed by an EARLIER launch of
            // this generation (whatever this launch writes is not read by it, so no fence orders an entry's stores)
            const bool usable = mine & ((prdy[p] & 0xFFull) == gen) & ((prdy[p] >> 8) < c.dyn_seq);
            const bool free_ = (ptag[p] & 0xFFull) != gen; // never used, or of another generation
            cand_p = usable ? p : cand_p;
            claimed |= mine & !usable;
            old_tag = free_ ? ptag[p] : old_tag;
            free_p = free_ ? p : free_p;
        }
        stamp(12);
        bool hit = false;
        const u64 *ent = c.dyn_memo + (size_t)((slot0 + (unsigned)(cand_p < 0 ? 0 : cand_p)) & (unsigned)(kMemoEntries - 1)) * kMemoStride;
        auto ld2 = [&](int i) -> ulonglong2 { return *reinterpret_cast<const ulonglong2 *>(ent + ME_VAL + i); };
        // the candidate's value header and ship records are requested with its key (one round trip for the common hit, not two)
        ulonglong2 vh = make_ulonglong2(0ull, 0ull);
        u64 sw[SSG_N_TRAFFIC][kMemoValShipWords];
        if (memo_ok & (cand_p >= 0)) {
            u64 diff = 0ull;
            // Header, ships and the first two arbiter slots of the key (slots past the env's count hold whatever an older entry
            // left and are masked out), and each section in ONE batch of 16-byte loads, all issued before
            // the first of them is looked at (the key's 21, then the value's header and ship records, 19).  (Written as a visitor that loads and compares pair by pair, the compiler issued
            // four loads, waited, compared, issued the next four: ten dependent round trips, 10-12 k cycles of every wave.)
            constexpr int kKb = 1 + 14 + 6, kVb = 1 + 3 * (kMemoValShipWords / 2);
            ulonglong2 kb[kKb], vb[kVb] = {};
            kb[0] = *reinterpret_cast<const ulonglong2 *>(ent + ME_KEY);
#pragma unroll
            for (int j = 0; j < 14; ++j) kb[1 + j] = *reinterpret_cast<const ulonglong2 *>(ent + ME_KEY + kMemoKeyShips + 2 * j);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int f = 0; f < 3; ++f) kb[15 + 3 * a + f] = *reinterpret_cast<const ulonglong2 *>(ent + ME_KEY + kMemoKeyArbs + kMemoArbWords * a + 2 * f);
#pragma unroll
            for (int i = 0; i < kKb; ++i) asm volatile("" : "+v"(kb[i].x), "+v"(kb[i].y));
            diff |= (kb[0].x ^ memo_hdr) | (kb[0].y ^ live0);
#pragma unroll
            for (int j = 0; j < 14; ++j) diff |= (kb[1 + j].x ^ key_ship_word(2 * j)) | (kb[1 + j].y ^ key_ship_word(2 * j + 1));
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                u64 da = 0ull;
#pragma unroll
                for (int f = 0; f < 3; ++f) da |= (kb[15 + 3 * a + f].x ^ key_arb_word(a, 2 * f)) | (kb[15 + 3 * a + f].y ^ key_arb_word(a, 2 * f + 1));
                diff |= (a < n_live) ? da : 0ull;
            }
            if ((incl != 0u) | (n_live > 2)) { // goals that take part and further arbiters, lane by lane
#pragma unroll
                for (int g = 0; g < SSG_MAX_GOALS; ++g) {
                    if (!((incl >> g) & 1u)) continue;
                    ulonglong2 gk[DC_GOAL_COLS / 2];
#pragma unroll
                    for (int f = 0; f < DC_GOAL_COLS / 2; ++f) gk[f] = *reinterpret_cast<const ulonglong2 *>(ent + ME_KEY + kMemoKeyGoals + DC_GOAL_COLS * g + 2 * f);
#pragma unroll
                    for (int f = 0; f < DC_GOAL_COLS / 2; ++f) diff |= (gk[f].x ^ dbits(gin[g][2 * f])) | (gk[f].y ^ dbits(gin[g][2 * f + 1]));
                }
#pragma unroll
                for (int a = 2; a < kMemoArbIn; ++a) {
                    if (a >= n_live) continue;
#pragma unroll
                    for (int f = 0; f < kMemoArbWords; f += 2) {
                        const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(ent + ME_KEY + kMemoKeyArbs + kMemoArbWords * a + f);
                        diff |= (w.x ^ key_arb_word(a, f)) | (w.y ^ key_arb_word(a, f + 1));
                    }
                }
            }
            // ... then, for the lanes whose key matched, the value's header and ship records: a second batch
            if (diff == 0ull) {
                vb[0] = ld2(0);
#pragma unroll
                for (int k = 0; k < SSG_N_TRAFFIC; ++k)
#pragma unroll
                    for (int i = 0; i < kMemoValShipWords / 2; ++i) vb[1 + (kMemoValShipWords / 2) * k + i] = ld2(kMemoValShips + kMemoValShipWords * k + 2 * i);
#pragma unroll
                for (int i = 0; i < kVb; ++i) asm volatile("" : "+v"(vb[i].x), "+v"(vb[i].y));
            }
            vh = vb[0];
#pragma unroll
            for (int k = 0; k < SSG_N_TRAFFIC; ++k)
#pragma unroll
                for (int i = 0; i < kMemoValShipWords / 2; ++i) { sw[k][2 * i] = vb[1 + (kMemoValShipWords / 2) * k + i].x; sw[k][2 * i + 1] = vb[1 + (kMemoValShipWords / 2) * k + i].y; }
            hit = diff == 0ull;
            // the value's counts index fixed-size sections of the entry and of this env's columns: an entry whose header is
            // not one this kernel could have written (the blob is the caller's) counts as a miss
            const unsigned vh0 = (unsigned)vh.x;
            hit &= ((vh0 >> 8) & 0xFFu) <= (unsigned)kMemoArbOut && ((vh0 >> 16) & 0xFFu) <= (unsigned)kMemoAged;
        }
        {   // statistics (how often the table answers), spread over slots, fire and forget
            const u64 mh = __ballot(hit), mm = __ballot(!hit);
            if (lane == __ffsll((long long)(mh | mm)) - 1) {
                u64 *st = c.dyn_memo_stats + (size_t)(blockIdx.x & (kMemoStatSlots - 1)) * kMemoStatWords;
                if (mh) atomicAdd(st + 0, (u64)__popcll(mh));
                if (mm) atomicAdd(st + 1, (u64)__popcll(mm));
            }
        }
        stamp(13);
        if (hit) {
            auto bd = [](u64 w) -> double { return __longlong_as_double((long long)w); };
            const unsigned vhdr = (unsigned)vh.x;
            const u64 live_out = vh.y;
            const bool v_changed = (vhdr & 1u) != 0u;
            const int n_out = (int)((vhdr >> 8) & 0xFFu), n_aged = (int)((vhdr >> 16) & 0xFFu);
            // the arbiter records and the participating goals: one more batch
            ulonglong2 ar[kMemoArbOut][3];
#pragma unroll
            for (int i = 0; i < kMemoArbOut; ++i) {
                if (i < n_out) { ar[i][0] = ld2(kMemoValArbs + kMemoArbWords * i); ar[i][1] = ld2(kMemoValArbs + kMemoArbWords * i + 2); ar[i][2] = ld2(kMemoValArbs + kMemoArbWords * i + 4); }
                if (i >= 2) break; // (more than two arbiters: fetched where they are stored, below)
            }
            // ships: 9 fields, cos, sin each; stored exactly as the computed step stores them (only a body that moved, or a fresh env)
#pragma unroll
            for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
                double *t = col.f64 + (size_t)(DC_TRAFFIC + 9 * k) * np + e;
                double *row = c.dyn_row + (size_t)e * kDynRow + kDynRowTraffic + 9 * k;
                bool dfb = false;
#pragma unroll
                for (int f = 0; f < 9; ++f) dfb |= sw[k][f] != dbits(tin[k][f]);
                if (dfb | fresh) {
#pragma unroll
                    for (int f = 0; f < 9; ++f) { t[(size_t)f * np] = bd(sw[k][f]); row[f] = bd(sw[k][f]); }
                    col.f64[(size_t)(DC_TROT + 2 * k) * np + e] = bd(sw[k][9]);
                    col.f64[(size_t)(DC_TROT + 2 * k + 1) * np + e] = bd(sw[k][10]);
                }
            }
#pragma unroll
            for (int g = 0; g < SSG_MAX_GOALS; ++g) {
                if (g >= ng || !((gmask >> g) & 1u)) continue;
                double *q = col.f64 + (size_t)(DC_GOALS + DC_GOAL_COLS * g) * np + e;
                double *row = c.dyn_row + (size_t)e * kDynRow + DC_GOAL_COLS * g;
                if ((incl >> g) & 1u) {
                    u64 gw[DC_GOAL_COLS];
#pragma unroll
                    for (int f = 0; f < DC_GOAL_COLS; f += 2) { const ulonglong2 w = ld2(kMemoValGoals + DC_GOAL_COLS * g + f); gw[f] = w.x; gw[f + 1] = w.y; }
                    bool dfb = false;
#pragma unroll
                    for (int f = 0; f < DC_GOAL_COLS; ++f) dfb |= gw[f] != dbits(gin[g][f]);
                    if (dfb | fresh) {
#pragma unroll
                        for (int f = 0; f < DC_GOAL_COLS; ++f) { q[(size_t)f * np] = bd(gw[f]); row[f] = bd(gw[f]); }
                    }
                } else if (fresh) { // a goal of a rebuilt env that takes no part: its fresh body, as the computed step stores it
#pragma unroll
                    for (int f = 0; f < DC_GOAL_COLS; ++f) { q[(size_t)f * np] = gin[g][f]; row[f] = gin[g][f]; }
                }
            }
            for (int i = 0; i < n_out; ++i) { // the arbiters the step left on the solver's list
                ulonglong2 a0, a1, a2;
                if (i == 0) { a0 = ar[0][0]; a1 = ar[0][1]; a2 = ar[0][2]; }
                else if (i == 1) { a0 = ar[1][0]; a1 = ar[1][1]; a2 = ar[1][2]; }
                else { a0 = ld2(kMemoValArbs + kMemoArbWords * i); a1 = ld2(kMemoValArbs + kMemoArbWords * i + 2); a2 = ld2(kMemoValArbs + kMemoArbWords * i + 4); }
                const u64 pk = a0.x;
                const int pid = (int)(pk & 0xFFull);
                if (pid >= kDynPairs) continue; // (never written by this kernel; see the header check above)
                col.u32[(size_t)(DU_META + pid) * np + e] = (unsigned)((pk >> 8) & 0xFFull);
                if (pid < kPolyPairs) col.u32[(size_t)(DU_HASH + pid) * np + e] = (unsigned)(pk >> 32);
                double *acc = col.f64 + (size_t)(DC_ARB + 4 * pid) * np + e;
                acc[0 * np] = bd(a0.y); acc[1 * np] = bd(a1.x); acc[2 * np] = bd(a1.y); acc[3 * np] = bd(a2.x);
            }
            for (int i = 0; i < n_aged; ++i) { // cached arbiters that aged this step
                const u64 pk = ent[ME_VAL + kMemoValAged + i];
                if ((int)(pk & 0xFFull) >= kDynPairs) continue;
                col.u32[(size_t)(DU_META + (int)(pk & 0xFFull)) * np + e] = (unsigned)((pk >> 8) & 0xFFull);
            }
            col.live[e] = live_out;
            c.dyn_hash[e] = (unsigned long long)d.bank_epoch;
            col.flag[e] = (uint8_t)(v_changed ? 0u : 4u);
            return;
        }
        // a miss: compute, then store — unless the state is not memoisable, somebody else is already storing it, or the probe
        // sequence has no free entry.  The entry is claimed NOW with one returning CAS whose answer is first looked at where
        // the result is stored (it travels while the narrowphase runs).
        memo_try = memo_ok & (cand_p < 0) & !claimed & (free_p >= 0);
        memo_ent = c.dyn_memo + (size_t)((slot0 + (unsigned)(free_p < 0 ? 0 : free_p)) & (unsigned)(kMemoEntries - 1)) * kMemoStride;
        if (memo_try) memo_old = atomicCAS(memo_ent + ME_TAG, old_tag, tag) ^ old_tag; // 0 = the entry is this lane's
    }
    stamp(11);
    // ---- narrowphase (cpCollide) of the surviving pairs, canonical order per env; one code site per pair type -------------
    Info info;
    for (;;) {
        // the wave's next pair = the lowest candidate of any lane (wave-uniform: one code site per trip; every lane still
        // meets its own pairs in ascending, i.e. canonical, order)
        // (minimum over the ACTIVE lanes by ballots, most significant bit first: lanes of a partly filled wave have left the
        // kernel, and a shuffle would read whatever their registers hold)
        const int ol = cand ? __ffsll((long long)cand) - 1 : 63;
        unsigned long long sel = __ballot(cand != 0ull);
        if (sel == 0ull) break;
#pragma unroll
        for (int b = 5; b >= 0; --b) {
            const unsigned long long zeros = __ballot((bool)((sel >> lane) & 1ull) & !((ol >> b) & 1));
            sel = zeros ? zeros : sel;
        }
        const int o = __builtin_amdgcn_readlane(ol, __ffsll((long long)sel) - 1);
        SSG_TYPE_BEGIN();
        int type_ = 0;
        if ((cand >> o) & 1ull) {
            cand &= ~(1ull << o);
            if (o < kGoalBlock) {
                const int g = (o >= o_goal(5)) ? 5 : (o >= o_goal(4)) ? 4 : (o >= o_goal(3)) ? 3 : (o >= o_goal(2)) ? 2 : (o >= o_goal(1)) ? 1 : 0;
                const int r = o - (2 * g + g * (g - 1) / 2);
                const CircleShape cg = goal_shape(g);
                if (r < 2) {
                    const BankShape<UNI> bs = bank_shape(r);
                    collide(cg, bs, info, emem);
                    push(info, g, slot_static, pid_gb(g, r), 0.0);
                    type_ = 0;
                } else {
                    type_ = 1;
                    const int h = r - 2;
                    const CircleShape ch = goal_shape(h);
                    collide(ch, cg, info, emem);
                    push(info, h, g, pid_gg(h, g), 0.0);
                }
            } else {
                const int k = (o >= o_ship(2)) ? 2 : (o >= o_ship(1)) ? 1 : 0;
                const int r = o - (kGoalBlock + (2 + SSG_MAX_GOALS) * k + k * (k - 1) / 2);
                const ShipShape sk = ship_shape(k);
                if (r < 2) {
                    // the narrowphase memo: this (record, side, ship, pose) may have been collided before (shipsim_internal.h, kNpmEntries)
                    bool got = false;
                    u64 *np_ent = nullptr;
                    u64 np_tag = 0ull, np_old = 0ull;
                    u64 np_hdr = 0ull;
                    const u64 gen_np = (u64)(c.dyn_memo_gen & 0xFFu);
                    if constexpr (MEMO) {
                        np_hdr = (u64)((unsigned)map_id & 0xFFFFu) | ((u64)(unsigned)r << 16) | ((u64)(unsigned)k << 20) | ((u64)(d.memo_fp & 0xFFFFu) << 24) |
                                 ((u64)(d.bank_epoch & 0xFFFFFFu) << 40);
                        u64 h0 = np_hdr ^ 0x452821E638D01377ull, h1 = 0xBE5466CF34E90C6Cull;
                        h0 = ((h0 << 7) | (h0 >> 57)) ^ dbits(sk.p.x); h1 = ((h1 << 11) | (h1 >> 53)) ^ dbits(sk.p.y);
                        h0 = ((h0 << 7) | (h0 >> 57)) ^ dbits(sk.ca);  h1 = ((h1 << 11) | (h1 >> 53)) ^ dbits(sk.sa);
                        const u64 hh = mix(mix(h0, h1), h0 >> 32);
                        const u64 gen = (u64)(c.dyn_memo_gen & 0xFFu);
                        np_tag = (hh & ~0xFFull) | gen;
                        const unsigned slot0 = (unsigned)(hh >> 24);
                        int cp = -1, fp_ = -1;
                        bool claimed = false;
#pragma unroll
                        for (int p = kNpmProbes - 1; p >= 0; --p) {
                            const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(c.dyn_npm + (size_t)((slot0 + (unsigned)p) & (unsigned)(kNpmEntries - 1)) * kNpmStride);
                            const bool mine = a.x == np_tag;
                            const bool usable = mine & ((a.y & 0xFFull) == gen) & ((a.y >> 8) < c.dyn_seq);
                            const bool free_ = (a.x & 0xFFull) != gen;
                            cp = usable ? p : cp;
                            claimed |= mine & !usable;
                            np_old = free_ ? a.x : np_old;
                            fp_ = free_ ? p : fp_;
                        }
                        if (cp >= 0) {
                            const u64 *ent = c.dyn_npm + (size_t)((slot0 + (unsigned)cp) & (unsigned)(kNpmEntries - 1)) * kNpmStride;
                            const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(ent + NE_KEY);
                            const ulonglong2 k0 = q[0], k1 = q[1], k2 = q[2];
                            const ulonglong2 v0 = q[3], v1 = q[4], v2 = q[5], v3 = q[6], v4 = q[7], v5 = q[8];
                            got = (((k0.x ^ np_hdr) | (k0.y ^ dbits(sk.p.x)) | (k1.x ^ dbits(sk.p.y)) | (k1.y ^ dbits(sk.ca)) | (k2.x ^ dbits(sk.sa))) == 0ull);
                            if (got) {
                                auto bd = [](u64 w) -> double { return __longlong_as_double((long long)w); };
                                info.count = (int)(v0.x & 0xFFull);
                                info.hash[0] = (unsigned)((v0.x >> 8) & 0xFFFFFFull); info.hash[1] = (unsigned)(v0.x >> 32);
                                info.n = mk(bd(v0.y), bd(v1.x));
                                info.p1[0] = mk(bd(v1.y), bd(v2.x)); info.p2[0] = mk(bd(v2.y), bd(v3.x));
                                info.p1[1] = mk(bd(v3.y), bd(v4.x)); info.p2[1] = mk(bd(v4.y), bd(v5.x));
                            }
                        }
                        if (!got & (cp < 0) & !claimed & (fp_ >= 0))
                            np_ent = c.dyn_npm + (size_t)((slot0 + (unsigned)fp_) & (unsigned)(kNpmEntries - 1)) * kNpmStride;
                    }
                    if (!got) {
                        const BankShape<UNI> bs = bank_shape(r);
                        collide(sk, bs, info, emem);
                        if constexpr (MEMO) {
                            if (np_ent && atomicCAS(np_ent + ME_TAG, np_old, np_tag) == np_old) {
                                np_ent[NE_KEY + 0] = np_hdr; np_ent[NE_KEY + 1] = dbits(sk.p.x); np_ent[NE_KEY + 2] = dbits(sk.p.y);
                                np_ent[NE_KEY + 3] = dbits(sk.ca); np_ent[NE_KEY + 4] = dbits(sk.sa); np_ent[NE_KEY + 5] = 0ull;
                                const bool c1 = info.count > 0, c2 = info.count > 1;
                                np_ent[NE_VAL + 0] = (u64)(unsigned)info.count | ((u64)(c1 ? info.hash[0] : 0u) << 8) | ((u64)(c2 ? info.hash[1] : 0u) << 32);
                                np_ent[NE_VAL + 1] = dbits(info.n.x); np_ent[NE_VAL + 2] = dbits(info.n.y);
                                np_ent[NE_VAL + 3] = dbits(c1 ? info.p1[0].x : 0.0); np_ent[NE_VAL + 4] = dbits(c1 ? info.p1[0].y : 0.0);
                                np_ent[NE_VAL + 5] = dbits(c1 ? info.p2[0].x : 0.0); np_ent[NE_VAL + 6] = dbits(c1 ? info.p2[0].y : 0.0);
                                np_ent[NE_VAL + 7] = dbits(c2 ? info.p1[1].x : 0.0); np_ent[NE_VAL + 8] = dbits(c2 ? info.p1[1].y : 0.0);
                                np_ent[NE_VAL + 9] = dbits(c2 ? info.p2[1].x : 0.0); np_ent[NE_VAL + 10] = dbits(c2 ? info.p2[1].y : 0.0);
                                np_ent[NE_VAL + 11] = 0ull;
                                np_ent[ME_READY] = (c.dyn_seq << 8) | gen_np;
                            }
                        }
                    }
                    if constexpr (MEMO) {
                        const u64 mg = __ballot(got), mn = __ballot(!got);
                        if (lane == __ffsll((long long)(mg | mn)) - 1) {
                            u64 *st = c.dyn_memo_stats + (size_t)(blockIdx.x & (kMemoStatSlots - 1)) * kMemoStatWords;
                            if (mg) atomicAdd(st + 3, (u64)__popcll(mg));
                            if (mn) atomicAdd(st + 4, (u64)__popcll(mn));
                        }
                    }
                    push(info, slot_ship0 + k, slot_static, pid_tb(k, r), d.ship_friction * 0.0);
                    type_ = 2;
                } else if (r < 2 + SSG_MAX_GOALS) {
                    type_ = 3;
                    const int g = r - 2;
                    const CircleShape cg = goal_shape(g);
                    collide(cg, sk, info, emem);
                    push(info, g, slot_ship0 + k, pid_gt(g, k), 0.0 * d.ship_friction);
                } else {
                    const int j = r - 2 - SSG_MAX_GOALS;
                    const ShipShape sj = ship_shape(j);
                    collide(sj, sk, info, emem);
                    push(info, slot_ship0 + j, slot_ship0 + k, pid_tt(j, k), d.ship_friction * d.ship_friction);
                    type_ = 4;
                }
            }
        }
        { // (the trip's pair type is wave-uniform: o is) gb gg tb gt tt
            int tt_;
            if (o < kGoalBlock) {
                const int g_ = (o >= o_goal(5)) ? 5 : (o >= o_goal(4)) ? 4 : (o >= o_goal(3)) ? 3 : (o >= o_goal(2)) ? 2 : (o >= o_goal(1)) ? 1 : 0;
                tt_ = (o - o_goal(g_) < 2) ? 0 : 1;
            } else {
                const int k_ = (o >= o_ship(2)) ? 2 : (o >= o_ship(1)) ? 1 : 0;
                const int r_ = o - o_ship(k_);
                tt_ = r_ < 2 ? 2 : (r_ < 2 + SSG_MAX_GOALS ? 3 : 4);
            }
            (void)tt_; (void)type_;
            SSG_TYPE_END(tt_);
        }
    }

    SSG_TICK(emem, 0);
#ifdef SSG_DYN_PROFILE
    if (d.stop_after == -1)
        for (int i = 0; i < 6; ++i) col.f64[(size_t)(DC_ARB + 4 * 45 + i) * np + e] = (double)prof_acc[i]; // unused arbiter rows of pairs 45, 46
    if (d.stop_after == -1)
        for (int i = 0; i < 5; ++i) col.f64[(size_t)(DC_ARB + 4 * 47 + i) * np + e] = (double)type_acc[i]; // ... and of pairs 47, 48
    if (d.stop_after == -1)
        for (int i = 0; i < 5; ++i) col.f64[(size_t)(DC_ARB + 210 + i) * np + e] = type_cnt[i]; // (pairs 52, 53: beyond the stamps' rows)
#endif
    stamp(2);
    if (d.stop_after == 3) return;
    // ---- cpSpaceArbiterSetFilter for the cached arbiters that were not touched this step -------------------------
    {
        unsigned long long rest = live & ~touched;
        while (rest) {
            const int pid = __ffsll((long long)rest) - 1;
            rest &= rest - 1ull;
            unsigned meta = 0u, hh_ = 0u;
            double a0_, a1_, a2_, a3_;
            arb_old(pid, meta, hh_, a0_, a1_, a2_, a3_);
            unsigned age = (meta >> 3) & 3u;
            age += 1u; // ticks >= 1: the arbiter is (now) "cached"
            changed = true; // an ageing arbiter is a state change by itself
            if (age >= (unsigned)kPersist) {
                live &= ~(1ull << pid);
            } else {
                meta = (meta & ~0x1Fu) | (unsigned)ST_CACHED | (age << 3);
                col.u32[(size_t)(DU_META + pid) * np + e] = meta;
                if constexpr (MEMO) {
                    if (memo_n_aged < kMemoAged) memo_aged[memo_n_aged] = (unsigned long long)(unsigned)pid | ((unsigned long long)meta << 8);
                    memo_n_aged++;
                }
            }
        }
    }

    stamp(14);
    auto ints_of = [&](const auto &A, int &pid, int &a, int &b, int &count, int &state) {
        const unsigned v = (unsigned)__double_as_longlong(A.get(A_INTS));
        pid = v & 0xFF; a = (v >> 8) & 0xFF; b = (v >> 16) & 0xFF; count = (v >> 24) & 0xF; state = v >> 28;
    };
    auto k_scalar_body = [&](int slot, V2 r, V2 n) -> double {
        const double rcn = cross(r, n);
        return m_inv_of(slot) + i_inv_of(slot) * rcn * rcn;
    };
    auto vel = [&](int slot) -> V2 { return mk(BF(slot, B_VX), BF(slot, B_VY)); };
    auto apply_impulse = [&](int slot, V2 j, V2 r) {
        const V2 v = vel(slot) + j * m_inv_of(slot);
        BF(slot, B_VX) = v.x; BF(slot, B_VY) = v.y;
        BF(slot, B_W) += i_inv_of(slot) * cross(r, j);
    };

    // (from here to the arbiters' write-back every access to an arbiter record goes through getA: the LDS-only form when no
    // lane of the wave has more than kLdsArb arbiters)
    auto solve_and_store = [&](auto getA, const bool memo_ins) {
    // ---- cpArbiterPreStep ---------------------------------------------------------------------------------------
    for (int i = 0; i < n_act; ++i) {
        const auto A = getA(i);
        int pid, a, b, count, state;
        ints_of(A, pid, a, b, count, state);
        const V2 n = mk(A.get(A_NX), A.get(A_NY));
        const V2 body_delta = mk(BF(b, B_PX), BF(b, B_PY)) - mk(BF(a, B_PX), BF(a, B_PY));
        for (int k = 0; k < count; ++k) {
            const V2 r1 = mk(A.cget(k, AC_R1X), A.cget(k, AC_R1Y)), r2 = mk(A.cget(k, AC_R2X), A.cget(k, AC_R2Y));
            A.cset(k, AC_NMASS, 1.0 / (k_scalar_body(a, r1, n) + k_scalar_body(b, r2, n)));
            A.cset(k, AC_TMASS, 1.0 / (k_scalar_body(a, r1, perp(n)) + k_scalar_body(b, r2, perp(n))));
            const double dist = dot((r2 - r1) + body_delta, n);
            A.cset(k, AC_BIAS, -d.bias_coef * cmin(0.0, dist + d.slop) / dt);
            A.cset(k, AC_JBIAS, 0.0);
            // con->bounce = normal_relative_velocity * e with e = 0 for every shape on this path: +-0, carried as 0
        }
    }
    stamp(3);
    if (d.stop_after == 4) return;
    // ---- (4) cpBodyUpdateVelocity (no forces on these bodies) -------------------------------------------------------
    for (int s = 0; s < slot_static; ++s) {
        if (s < ng && !((gmask >> s) & 1u)) continue; // goal no longer in the space
        const V2 v = vel(s) * c.damp + (mk(0, 0) + mk(0, 0) * m_inv_of(s)) * dt;
        BF(s, B_VX) = v.x; BF(s, B_VY) = v.y;
        BF(s, B_W) = BF(s, B_W) * c.damp + 0.0 * i_inv_of(s) * dt;
    }
    // ---- (5) cached impulses (dt_coef = dt/prev_dt = 1; first contacts skip), then the solver ----------------------
    for (int i = 0; i < n_act; ++i) {
        const auto A = getA(i);
        int pid, a, b, count, state;
        ints_of(A, pid, a, b, count, state);
        if (state == ST_FIRST) continue;
        const V2 n = mk(A.get(A_NX), A.get(A_NY));
        for (int k = 0; k < count; ++k) {
            const V2 r1 = mk(A.cget(k, AC_R1X), A.cget(k, AC_R1Y)), r2 = mk(A.cget(k, AC_R2X), A.cget(k, AC_R2Y));
            const V2 j = rotate(n, mk(A.cget(k, AC_JN), A.cget(k, AC_JT))) * 1.0;
            apply_impulse(a, neg(j), r1);
            apply_impulse(b, j, r2);
        }
    }
    // cpSpaceStep's 10 solver iterations.  Almost every env has one or two arbiters on its list: those run entirely in
    // registers (the records' constants, the accumulated impulses and both bodies' velocity fields are loaded once, the
    // two arbiters hand a shared body's fields to each other after each update, everything is written back once); an
    // iteration is then ~60 dependent FP64 operations instead of that plus ~40 LDS round trips with computed addresses.
    // Longer lists take the generic LDS-resident loop below.
    struct RegArb {
        int a, b, count;
        V2 n;
        double u, ma, ia, mb, ib;
        V2 r1[2], r2[2];
        double nM[2], tM[2], bias[2], jB[2], jn[2], jt[2];
        V2 av, avb, bv, bvb;
        double aw, awb, bw, bwb;
    };
    auto reg_load = [&](RegArb &R, int i) {
        const auto A = getA(i);
        int pid, state;
        ints_of(A, pid, R.a, R.b, R.count, state);
        R.n = mk(A.get(A_NX), A.get(A_NY));
        R.u = A.get(A_U);
        R.ma = m_inv_of(R.a); R.ia = i_inv_of(R.a); R.mb = m_inv_of(R.b); R.ib = i_inv_of(R.b);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            R.r1[k] = mk(A.cget(k, AC_R1X), A.cget(k, AC_R1Y)); R.r2[k] = mk(A.cget(k, AC_R2X), A.cget(k, AC_R2Y));
            R.nM[k] = A.cget(k, AC_NMASS); R.tM[k] = A.cget(k, AC_TMASS); R.bias[k] = A.cget(k, AC_BIAS);
            R.jB[k] = A.cget(k, AC_JBIAS); R.jn[k] = A.cget(k, AC_JN); R.jt[k] = A.cget(k, AC_JT);
        }
        R.av = vel(R.a); R.avb = mk(BF(R.a, B_VBX), BF(R.a, B_VBY)); R.aw = BF(R.a, B_W); R.awb = BF(R.a, B_WB);
        R.bv = vel(R.b); R.bvb = mk(BF(R.b, B_VBX), BF(R.b, B_VBY)); R.bw = BF(R.b, B_W); R.bwb = BF(R.b, B_WB);
    };
    auto reg_step = [&](RegArb &R) { // cpArbiterApplyImpulse
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= R.count) break;
            const V2 r1 = R.r1[k], r2 = R.r2[k], n = R.n;
            const V2 vb1 = R.avb + perp(r1) * R.awb;
            const V2 vb2 = R.bvb + perp(r2) * R.bwb;
            const V2 v1 = R.av + perp(r1) * R.aw;
            const V2 v2 = R.bv + perp(r2) * R.bw;
            const V2 vr = (v2 - v1) + mk(0, 0);
            const double vbn = dot(vb2 - vb1, n);
            const double vrn = dot(vr, n);
            const double vrt = dot(vr, perp(n));
            const double jbn = (R.bias[k] - vbn) * R.nM[k];
            const double jbnOld = R.jB[k];
            const double jBias = cmax(jbnOld + jbn, 0.0);
            const double jn = -(0.0 + vrn) * R.nM[k];
            const double jnOld = R.jn[k];
            const double jnAcc = cmax(jnOld + jn, 0.0);
            const double jtMax = R.u * jnAcc;
            const double jt = -vrt * R.tM[k];
            const double jtOld = R.jt[k];
            const double jtAcc = cclamp(jtOld + jt, -jtMax, jtMax);
            R.jB[k] = jBias; R.jn[k] = jnAcc; R.jt[k] = jtAcc;
            const V2 jb = n * (jBias - jbnOld);
            const V2 j = rotate(n, mk(jnAcc - jnOld, jtAcc - jtOld));
            const V2 njb = neg(jb), nj = neg(j);
            R.avb = R.avb + njb * R.ma; R.bvb = R.bvb + jb * R.mb;         // apply_bias_impulses
            R.awb = R.awb + R.ia * cross(r1, njb); R.bwb = R.bwb + R.ib * cross(r2, jb);
            R.av = R.av + nj * R.ma; R.bv = R.bv + j * R.mb;               // apply_impulses
            R.aw = R.aw + R.ia * cross(r1, nj); R.bw = R.bw + R.ib * cross(r2, j);
        }
    };
    auto reg_sync = [&](const RegArb &F, RegArb &T) { // T's copy of a body F has just updated
        if (T.a == F.a) { T.av = F.av; T.avb = F.avb; T.aw = F.aw; T.awb = F.awb; }
        if (T.a == F.b) { T.av = F.bv; T.avb = F.bvb; T.aw = F.bw; T.awb = F.bwb; }
        if (T.b == F.a) { T.bv = F.av; T.bvb = F.avb; T.bw = F.aw; T.bwb = F.awb; }
        if (T.b == F.b) { T.bv = F.bv; T.bvb = F.bvb; T.bw = F.bw; T.bwb = F.bwb; }
    };
    auto reg_store = [&](const RegArb &R, int i) {
        const auto A = getA(i);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= R.count) break;
            A.cset(k, AC_JBIAS, R.jB[k]); A.cset(k, AC_JN, R.jn[k]); A.cset(k, AC_JT, R.jt[k]);
        }
        BF(R.a, B_VX) = R.av.x; BF(R.a, B_VY) = R.av.y; BF(R.a, B_W) = R.aw;
        BF(R.a, B_VBX) = R.avb.x; BF(R.a, B_VBY) = R.avb.y; BF(R.a, B_WB) = R.awb;
        BF(R.b, B_VX) = R.bv.x; BF(R.b, B_VY) = R.bv.y; BF(R.b, B_W) = R.bw;
        BF(R.b, B_VBX) = R.bvb.x; BF(R.b, B_VBY) = R.bvb.y; BF(R.b, B_WB) = R.bwb;
    };
    if (n_act >= 1 && n_act <= 2) {
        RegArb R0, R1;
        reg_load(R0, 0);
        reg_load(R1, n_act > 1 ? 1 : 0);
        const bool two = n_act > 1;
        // the hand-over of a shared body between the two arbiters (~200 selects per iteration) is only needed if some lane of
        // the wave HAS a shared dynamic body (the static body's fields never change: its inverse mass and moment are 0)
        const bool shares = two & ((((R0.a == R1.a) | (R0.a == R1.b)) & (R0.a != slot_static)) | (((R0.b == R1.a) | (R0.b == R1.b)) & (R0.b != slot_static)));
        const bool any_shared = __any(shares);
        for (int it = 0; it < kIter; ++it) {
            reg_step(R0);
            if (two) {
                if (any_shared) reg_sync(R0, R1);
                reg_step(R1);
                if (any_shared) reg_sync(R1, R0);
            }
        }
        reg_store(R0, 0);
        if (two) reg_store(R1, 1); // (a body the two arbiters share holds the same, latest fields in both after the last sync)
    } else {
        for (int it = 0; it < kIter; ++it) {
            for (int i = 0; i < n_act; ++i) {
                const auto A = getA(i);
                int pid, a, b, count, state;
                ints_of(A, pid, a, b, count, state);
                const V2 n = mk(A.get(A_NX), A.get(A_NY));
                const double u = A.get(A_U);
                const double ma = m_inv_of(a), ia = i_inv_of(a), mb = m_inv_of(b), ib = i_inv_of(b);
                for (int k = 0; k < count; ++k) {
                    // one batch of LDS reads (the record's contact fields, both bodies' velocity fields), the arithmetic of
                    // cpArbiterApplyImpulse, one batch of writes: a and b are different bodies, so the four apply_*impulse
                    // updates need not re-read what the previous one wrote
                    const V2 r1 = mk(A.cget(k, AC_R1X), A.cget(k, AC_R1Y)), r2 = mk(A.cget(k, AC_R2X), A.cget(k, AC_R2Y));
                    const double nMass = A.cget(k, AC_NMASS), tMass = A.cget(k, AC_TMASS), bias = A.cget(k, AC_BIAS);
                    const double jbnOld = A.cget(k, AC_JBIAS), jnOld = A.cget(k, AC_JN), jtOld = A.cget(k, AC_JT);
                    const V2 a_v = vel(a), b_v = vel(b), a_vb = mk(BF(a, B_VBX), BF(a, B_VBY)), b_vb = mk(BF(b, B_VBX), BF(b, B_VBY));
                    const double a_w = BF(a, B_W), b_w = BF(b, B_W), a_wb = BF(a, B_WB), b_wb = BF(b, B_WB);
                    const V2 vb1 = a_vb + perp(r1) * a_wb;
                    const V2 vb2 = b_vb + perp(r2) * b_wb;
                    const V2 v1 = a_v + perp(r1) * a_w;
                    const V2 v2 = b_v + perp(r2) * b_w;
                    const V2 vr = (v2 - v1) + mk(0, 0);
                    const double vbn = dot(vb2 - vb1, n);
                    const double vrn = dot(vr, n);
                    const double vrt = dot(vr, perp(n));
                    const double jbn = (bias - vbn) * nMass;
                    const double jBias = cmax(jbnOld + jbn, 0.0);
                    const double jn = -(0.0 + vrn) * nMass;
                    const double jnAcc = cmax(jnOld + jn, 0.0);
                    const double jtMax = u * jnAcc;
                    const double jt = -vrt * tMass;
                    const double jtAcc = cclamp(jtOld + jt, -jtMax, jtMax);
                    A.cset(k, AC_JBIAS, jBias); A.cset(k, AC_JN, jnAcc); A.cset(k, AC_JT, jtAcc);
                    const V2 jb = n * (jBias - jbnOld);
                    const V2 j = rotate(n, mk(jnAcc - jnOld, jtAcc - jtOld));
                    const V2 njb = neg(jb), nj = neg(j);
                    const V2 a_vb2 = a_vb + njb * ma, b_vb2 = b_vb + jb * mb; // apply_bias_impulses
                    const double a_wb2 = a_wb + ia * cross(r1, njb), b_wb2 = b_wb + ib * cross(r2, jb);
                    const V2 a_v2 = a_v + nj * ma, b_v2 = b_v + j * mb;       // apply_impulses
                    const double a_w2 = a_w + ia * cross(r1, nj), b_w2 = b_w + ib * cross(r2, j);
                    BF(a, B_VBX) = a_vb2.x; BF(a, B_VBY) = a_vb2.y; BF(a, B_WB) = a_wb2;
                    BF(a, B_VX) = a_v2.x; BF(a, B_VY) = a_v2.y; BF(a, B_W) = a_w2;
                    BF(b, B_VBX) = b_vb2.x; BF(b, B_VBY) = b_vb2.y; BF(b, B_WB) = b_wb2;
                    BF(b, B_VX) = b_v2.x; BF(b, B_VY) = b_v2.y; BF(b, B_W) = b_w2;
                }
            }
        }
    }

    stamp(4);
    if (d.stop_after == 5) return;
    // ---- write back, hashing what is written --------------------------------------------------------------------------
    for (int i = 0; i < n_act; ++i) {
        const auto A = getA(i);
        int pid, a, b, count, state;
        ints_of(A, pid, a, b, count, state);
        if (!((live0 >> pid) & 1ull)) changed = true; // a new arbiter
        const unsigned meta = (unsigned)state | (0u << 3) | ((unsigned)count << 5);
        const unsigned long long hh64 = (unsigned long long)__double_as_longlong(A.get(A_HASH));
        const unsigned hh = (pid < kPolyPairs) ? (((unsigned)hh64 & 0xFFFFu) | ((count > 1 ? (unsigned)(hh64 >> 32) : 0u) << 16)) : 0u;
        const double j0 = A.cget(0, AC_JN), j1 = count > 1 ? A.cget(1, AC_JN) : 0.0;
        const double t0 = A.cget(0, AC_JT), t1 = count > 1 ? A.cget(1, AC_JT) : 0.0;
        aout ^= arb_hash(pid, meta, hh, j0, j1, t0, t1);
        col.u32[(size_t)(DU_META + pid) * np + e] = meta;
        if (pid < kPolyPairs) col.u32[(size_t)(DU_HASH + pid) * np + e] = hh;
        double *acc = col.f64 + (size_t)(DC_ARB + 4 * pid) * np + e;
        acc[0 * np] = j0; acc[1 * np] = j1; acc[2 * np] = t0; acc[3 * np] = t1;
        if constexpr (MEMO) {
            if (memo_ins) {
                u64 *ar = memo_ent + ME_VAL + kMemoValArbs + kMemoArbWords * i;
                ar[0] = (u64)(unsigned)pid | ((u64)meta << 8) | ((u64)hh << 32);
                ar[1] = dbits(j0); ar[2] = dbits(j1); ar[3] = dbits(t0); ar[4] = dbits(t1); ar[5] = 0ull;
            }
        }
    }
    };
    // the memo: did this lane claim an entry (the CAS issued at the miss)?  Then the key — the INPUT state — goes in now, so that
    // the registers that hold it (the cached arbiters' records) are free during the solver, as in the kernel without a memo.
    bool memo_ins = false;
    if constexpr (MEMO) {
        asm volatile("" : "+v"(memo_old)); // (first use of the CAS's answer: not before this point)
        memo_ins = memo_try & (memo_old == 0ull) & (memo_n_aged <= kMemoAged); // (a claimed entry that is not filled stays unusable: rare)
        if (memo_ins) {
            key_visit([&](int i, u64 w0, u64 w1) { memo_ent[ME_KEY + i] = w0; memo_ent[ME_KEY + i + 1] = w1; }, memo_incl, memo_n_live);
#pragma unroll
            for (int i = 0; i < kMemoAged; ++i) memo_ent[ME_VAL + kMemoValAged + i] = memo_aged[i];
        }
    }
    if (!__any(n_act > kLdsArb)) solve_and_store(arb_lds, memo_ins); else solve_and_store(arb, memo_ins);
    auto differs = [](double a, double b) -> bool { return __double_as_longlong(a) != __double_as_longlong(b); };
#pragma unroll
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        double *t = col.f64 + (size_t)(DC_TRAFFIC + 9 * k) * np + e;
        const int s = slot_ship0 + k;
        const double v[9] = {BF(s, B_PX), BF(s, B_PY), L(xbase + X_STRIDE * k + X_A), BF(s, B_VX), BF(s, B_VY), BF(s, B_W),
                             BF(s, B_VBX), BF(s, B_VBY), BF(s, B_WB)};
        // (columns and row hold what was loaded: only the fields this step changed are stored — in the post-reset transient
        // that is ship 1 and whatever it shoves, not the 75 fields of the env)
        double *row = c.dyn_row + (size_t)e * kDynRow + kDynRowTraffic + 9 * k;
        // (one branch per BODY, not per field: a body either moved — nearly all of its fields differ — or it did not)
        bool dfb = false;
#pragma unroll
        for (int f = 0; f < 9; ++f) dfb |= differs(v[f], tin[k][f]);
        if (dfb | fresh) {
#pragma unroll
            for (int f = 0; f < 9; ++f) { t[(size_t)f * np] = v[f]; row[f] = v[f]; }
            // the rotation of the angle column (cos, sin as this step's cpBodyUpdatePosition computed them): what the step
            // kernel's collide_ship builds the ship's world hull from
            col.f64[(size_t)(DC_TROT + 2 * k) * np + e] = L(xbase + X_STRIDE * k + X_CA);
            col.f64[(size_t)(DC_TROT + 2 * k + 1) * np + e] = L(xbase + X_STRIDE * k + X_SA);
        }
        changed |= dfb;
        if constexpr (MEMO) {
            if (memo_ins) {
                u64 *sv = memo_ent + ME_VAL + kMemoValShips + kMemoValShipWords * k;
#pragma unroll
                for (int f = 0; f < 9; ++f) sv[f] = dbits(v[f]);
                sv[9] = dbits(L(xbase + X_STRIDE * k + X_CA)); sv[10] = dbits(L(xbase + X_STRIDE * k + X_SA)); sv[11] = 0ull;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < SSG_MAX_GOALS; ++g) {
        if (g >= ng || !((gmask >> g) & 1u)) continue;
        double *q = col.f64 + (size_t)(DC_GOALS + DC_GOAL_COLS * g) * np + e;
        const double v[DC_GOAL_COLS] = {BF(g, B_PX), BF(g, B_PY), BF(g, B_VX), BF(g, B_VY), BF(g, B_VBX), BF(g, B_VBY),
                                        BF(g, B_W), BF(g, B_WB)};
        double *row = c.dyn_row + (size_t)e * kDynRow + DC_GOAL_COLS * g;
        bool dfb = false;
#pragma unroll
        for (int f = 0; f < DC_GOAL_COLS; ++f) dfb |= differs(v[f], gin[g][f]);
        if (dfb | fresh) {
#pragma unroll
            for (int f = 0; f < DC_GOAL_COLS; ++f) { q[(size_t)f * np] = v[f]; row[f] = v[f]; }
        }
        changed |= dfb;
        if constexpr (MEMO) {
            if (memo_ins && ((memo_incl >> g) & 1u)) {
#pragma unroll
                for (int f = 0; f < DC_GOAL_COLS; ++f) memo_ent[ME_VAL + kMemoValGoals + DC_GOAL_COLS * g + f] = dbits(v[f]);
            }
        }
    }
    changed |= (live != live0) | (ain != aout);
    if constexpr (MEMO) {
        if (memo_ins) {
            // (a goal outside the key that this step nevertheless changed cannot exist: it had no velocity and no candidate pair)
            memo_ent[ME_VAL + 0] = (u64)(changed ? 1u : 0u) | ((u64)(unsigned)n_act << 8) | ((u64)(unsigned)memo_n_aged << 16);
            memo_ent[ME_VAL + 1] = live;
            memo_ent[ME_READY] = (c.dyn_seq << 8) | (u64)(c.dyn_memo_gen & 0xFFu); // (the entry's last store)
            atomicAdd(c.dyn_memo_stats + (size_t)(blockIdx.x & (kMemoStatSlots - 1)) * kMemoStatWords + 2, 1ull);
        }
    }
    stamp(5);
    if (d.stop_after == -1) {
        col.f64[(size_t)(DC_ARB + 4 * 50 + 6) * np + e] = (double)n_act;
        col.f64[(size_t)(DC_ARB + 4 * 50 + 7) * np + e] = (double)dbg_cnt[0];
        col.f64[(size_t)(DC_ARB + 4 * 50 + 8) * np + e] = (double)dbg_cnt[1];
        col.f64[(size_t)(DC_ARB + 4 * 50 + 9) * np + e] = (double)dbg_cnt[2];
    }
    col.live[e] = live;
    c.dyn_hash[e] = (unsigned long long)d.bank_epoch; // the bank generation this (possible) rest state belongs to
    col.flag[e] = (uint8_t)(changed ? 0u : 4u); // unchanged = a fixed point of cpSpaceStep: at rest
}

static size_t dyn_lds_bytes(int n_goals, bool uni) // (per translation unit: its kGrp)
{
    return ((size_t)dyn_lane_doubles(n_goals, uni) * kGrp + (size_t)kHullDoubles * (1 + SSG_N_TRAFFIC) + (uni ? 2 * kBankDoubles : 0)) * sizeof(double);
}

#ifdef SSG_DYN_NONUNI_TU
// The per-lane-planes variant (banks of more than 64 records, per-env rings of worlds: nothing tells a wave which record its lanes
// sit on, so every lane stages its own two hulls: 283 doubles of LDS per lane).  Compiled in a translation unit of its own with 32
// envs per wave: 72 KB per wave, two waves per CU, 16 384 envs resident at once — at 48 (108 KB, one wave per CU, 12 288 envs) the
// ~16 k queued envs of a steady-state step needed two rounds.
hipError_t prepare_dyn_nonuni(const DevCfg &c)
{
    if (dyn_lds_bytes(c.n_goals, false) > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dyn_step_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(dyn_step_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_dyn_step_nonuni(const DevCfg &c, const DynCfg &dd, hipStream_t stream)
{
    const dim3 grid((unsigned)((c.n_pad + kDynMapBuckets * kGrp) / kGrp));
    // (a SHARED bank of more than 64 records: the waves no longer sit on one record each, but the envs still replay each other's
    // states — the memo, keyed by the record, answers them just the same; per-env rings of worlds: nothing repeats, no memo)
    if (c.dyn_memo) hipLaunchKernelGGL((dyn_step_kernel<false, true>), grid, dim3(64), dyn_lds_bytes(c.n_goals, false), stream, c, dd);
    else hipLaunchKernelGGL((dyn_step_kernel<false, false>), grid, dim3(64), dyn_lds_bytes(c.n_goals, false), stream, c, dd);
    return hipGetLastError();
}
#else
hipError_t prepare_dyn_nonuni(const DevCfg &c);
hipError_t launch_dyn_step_nonuni(const DevCfg &c, const DynCfg &dd, hipStream_t stream);

// Raise the dynamic-LDS cap of the full-step kernel (once per handle, like prepare_step).
hipError_t prepare_dyn(const DevCfg &c)
{
    if (dyn_lds_bytes(c.n_goals, true) > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dyn_step_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(dyn_step_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return prepare_dyn_nonuni(c);
}

hipError_t launch_dyn_step(const DevCfg &c, const DynCfg &d, bool classify, hipStream_t stream)
{
    // (the classify pass over every env only when the host touched the envs,) then the full step over the bucketed queue (grid
    // sized for the worst case: every env queued and every map's stretch rounded up to a wave; workgroups past the queue's end
    // leave at once).
    static const int stop_after = [] { const char *sv = getenv("SSG_DYN_STOP"); return sv ? atoi(sv) : 0; }(); // dev aid
    DynCfg dd = d;
    dd.stop_after = stop_after;
    if (classify)
        hipLaunchKernelGGL(dyn_classify_kernel, dim3((unsigned)((c.n_pad + kClassifyThreads - 1) / kClassifyThreads)),
                           dim3(kClassifyThreads), 0, stream, c, dd);
    // one bank record per wave: the buckets tell records apart only when the bank holds at most kDynMapBuckets of them
    // (a per-env ring of worlds never does)
    const bool uni = c.map_ring == 0 && c.n_maps <= kDynMapBuckets;
    const dim3 grid((unsigned)((c.n_pad + kDynPad) / kGrp));
    if (uni && c.dyn_memo) hipLaunchKernelGGL((dyn_step_kernel<true, true>), grid, dim3(64), dyn_lds_bytes(c.n_goals, true), stream, c, dd);
    else if (uni) hipLaunchKernelGGL((dyn_step_kernel<true, false>), grid, dim3(64), dyn_lds_bytes(c.n_goals, true), stream, c, dd);
    else return launch_dyn_step_nonuni(c, dd, stream); // (its own translation unit: 32 envs per wave)
    return hipGetLastError();
}

__global__ void dyn_invalidate_kernel(const DevCfg c, const uint8_t *__restrict__ mask)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= c.n_envs) return;
    const size_t np = (size_t)c.n_pad;
    if (!mask || mask[e]) {
        c.dyn_flag[e] &= (uint8_t)~4u;
        // the caller wrote the body columns: bring the row-major shadow the full step loads from up to date
        double *row = c.dyn_row + (size_t)e * kDynRow;
        for (int i = 0; i < DC_GOAL_COLS * SSG_MAX_GOALS; ++i) row[i] = c.dyn_f64[(size_t)(DC_GOALS + i) * np + e];
        for (int i = 0; i < 9 * SSG_N_TRAFFIC; ++i) row[kDynRowTraffic + i] = c.dyn_f64[(size_t)(DC_TRAFFIC + i) * np + e];
    }
    // the rotations of the (possibly rewritten) angle columns — for EVERY env, masked or not: the step kernel's collide_ship turns
    // the traffic hulls with these, and a caller's mask may name fewer envs than it wrote
    for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
        double sa, ca;
        sincos_body(c.dyn_f64[(size_t)(DC_TRAFFIC + 9 * k + 2) * np + e], &sa, &ca);
        c.dyn_f64[(size_t)(DC_TROT + 2 * k) * np + e] = ca;
        c.dyn_f64[(size_t)(DC_TROT + 2 * k + 1) * np + e] = sa;
    }
}

hipError_t launch_dyn_invalidate(const DevCfg &c, const uint8_t *mask, hipStream_t stream)
{
    hipLaunchKernelGGL(dyn_invalidate_kernel, dim3((unsigned)((c.n_envs + 255) / 256)), dim3(256), 0, stream, c, mask);
    return hipGetLastError();
}

hipError_t launch_dyn_reset(const DevCfg &c, const DynCfg &d, const uint8_t *mask, const int32_t *map_ids, double *obs, bool append, hipStream_t stream)
{
    const int block = 256;
    hipLaunchKernelGGL(dyn_reset_kernel, dim3((unsigned)((c.n_envs + block - 1) / block)), dim3(block), 0, stream, c, d, mask, map_ids, obs, append ? 1 : 0);
    return hipGetLastError();
}

#endif // SSG_DYN_NONUNI_TU
} // namespace ssg
