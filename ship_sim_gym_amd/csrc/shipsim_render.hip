// shipsim_render.hip — `rgb_array` frames of single envs (debugging / videos, not a hot path).
//
// Replaces ShipGame.render + get_screen (game.py:133-138,197-229): the screen is cleared to (0, 0, 200); in debug mode
// pymunk's debug draw paints every shape of the space in insertion order with its colour — banks (139, 69, 19)
// models.py:181, goals green game.py:88, the player white game.py:275, traffic black game.py:284-286 — then one
// radius-10 circle per lidar beam at the beam's end point (red where it hit a bank, green where it did not,
// game.py:207-225); finally a yellow radius-10 circle at the player's position (game.py:227-229).  Screen y points
// down (`invert_p`, game.py:73-75); the buffer is laid out like pygame.surfarray.array3d: [x][y][rgb].
// One thread per pixel: filled convex shapes are half-plane tests against the same splitting planes the physics uses.
// Not pixel-identical to pygame's scan conversion or pymunk's outline style (neither is available to compare).
#include <hip/hip_runtime.h>

#include <cmath>

#include "shipsim_internal.h"

namespace ssg {
namespace {

__device__ __forceinline__ bool in_hull(const double *pl, int n, int stride, double x, double y)
{
    bool in = n > 0;
    for (int j = 0; j < n; ++j) {
        const double *q = pl + stride * j;
        in &= (q[2] * (x - q[0]) + q[3] * (y - q[1])) <= 0.0;
    }
    return in;
}

// world vertices / normals of a 5-vertex ship hull at pose (px, py, angle), then the point test
__device__ bool in_ship(const double *hull, const double *nrm, double px, double py, double ang, double x, double y)
{
    double sa, ca;
    sincos(ang, &sa, &ca);
    bool in = true;
    for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
        const double hx = hull[2 * i], hy = hull[2 * i + 1], nx = nrm[2 * i], ny = nrm[2 * i + 1];
        const double vx = ca * hx + (-sa) * hy + px, vy = sa * hx + ca * hy + py;
        const double wx = ca * nx + (-sa) * ny, wy = sa * nx + ca * ny;
        in &= (wx * (x - vx) + wy * (y - vy)) <= 0.0;
    }
    return in;
}

// cpPolyShapeSegmentQuery (radius 0) of the segment a->b against one hull: smallest entering alpha, or 2 = miss
__device__ double seg_hull(const double *pl, int n, double ax, double ay, double bx, double by)
{
    double best = 2.0;
    for (int j = 0; j < n; ++j) {
        const double *q = pl + SSG_PLANE_DOUBLES * j;
        const double nx = q[2], ny = q[3];
        const double an = ax * nx + ay * ny, bn = bx * nx + by * ny;
        const double d = an - q[4];
        if (d < 0.0) continue;
        const double t = d / fmax(an - bn, 2.2250738585072014e-308);
        if (t < 0.0 || t > 1.0) continue;
        const double hx = ax + (bx - ax) * t, hy = ay + (by - ay) * t;
        const double dt = nx * hy - ny * hx;
        const double *qp = pl + SSG_PLANE_DOUBLES * ((j - 1 + n) % n);
        const double dtmin = nx * qp[1] - ny * qp[0], dtmax = nx * q[1] - ny * q[0];
        if (dtmin <= dt && dt <= dtmax && t < best) best = t;
    }
    return best;
}

} // namespace

__global__ void render_kernel(const DevCfg c, const DynCfg d, int e, int width, int height, uint8_t *__restrict__ rgb,
                              unsigned flags)
{
    __shared__ double beam_x[SSG_MAX_BEAMS], beam_y[SSG_MAX_BEAMS];
    __shared__ int beam_hit[SSG_MAX_BEAMS];
    const size_t np = (size_t)c.n_pad;
    const double px = c.f64cols[(size_t)COL_X * np + e], py = c.f64cols[(size_t)COL_Y * np + e];
    const double ang = c.f64cols[(size_t)COL_A * np + e];
    const int map_id = c.i32cols[(size_t)ICOL_MAP * np + e];
    const double *rec = c.bank + (size_t)map_id * SSG_MAP_STRIDE;
    const unsigned gmask = c.mask[e];
    const bool debug = flags & 1u;
    if (debug && threadIdx.x < (unsigned)c.n_beams) {
        // LiDAR.query on the current pose (models.py:39-76): origin = position + half the world AABB extents
        double sa, ca;
        sincos(ang, &sa, &ca);
        double l = INFINITY, r = -INFINITY, b = INFINITY, t = -INFINITY;
        for (int i = 0; i < SSG_SHIP_VERTS; ++i) {
            const double vx = ca * c.hull[2 * i] + (-sa) * c.hull[2 * i + 1] + px, vy = sa * c.hull[2 * i] + ca * c.hull[2 * i + 1] + py;
            l = fmin(l, vx); r = fmax(r, vx); b = fmin(b, vy); t = fmax(t, vy);
        }
        const double ox = px + (r - l) / 2, oy = py + (t - b) / 2;
        const int i = threadIdx.x;
        const double dx = ca * c.beam_cos[i] - sa * c.beam_sin[i], dy = sa * c.beam_cos[i] + ca * c.beam_sin[i];
        const double ex = ox + c.lidar_dist * dx, ey = oy + c.lidar_dist * dy;
        double alpha = 2.0;
        for (int s = 0; s < 2 && alpha > 1.0; ++s) // first listed shape that reports a hit wins
            alpha = seg_hull(rec + SSG_MAP_OFF_PLANES + s * SSG_MAX_HULL * SSG_PLANE_DOUBLES, (int)rec[SSG_MAP_OFF_COUNTS + s], ox, oy, ex, ey);
        const double tt = alpha <= 1.0 ? alpha : 1.0;
        beam_x[i] = ox + (ex - ox) * tt; beam_y[i] = oy + (ey - oy) * tt; beam_hit[i] = alpha <= 1.0;
    }
    __syncthreads();
    const int sx = blockIdx.x * blockDim.x + threadIdx.x, sy = blockIdx.y;
    if (sx >= width) return;
    // pixel centre -> world point (screen y points down)
    const double x = (sx + 0.5) * (c.width / width), y = c.height - (sy + 0.5) * (c.height / height);
    uint8_t R = 0, G = 0, B = 200; // screen.fill((0, 0, 200))
    if (debug) {
        for (int s = 0; s < 2; ++s)
            if (in_hull(rec + SSG_MAP_OFF_PLANES + s * SSG_MAX_HULL * SSG_PLANE_DOUBLES, (int)rec[SSG_MAP_OFF_COUNTS + s],
                        SSG_PLANE_DOUBLES, x, y)) { R = 139; G = 69; B = 19; }
        for (int g = 0; g < c.n_goals; ++g) {
            if (!((gmask >> g) & 1u)) continue;
            double gx, gy;
            if (c.n_ships > 1) { gx = c.dyn_f64[(size_t)(DC_GOALS + DC_GOAL_COLS * g) * np + e]; gy = c.dyn_f64[(size_t)(DC_GOALS + DC_GOAL_COLS * g + 1) * np + e]; }
            else { gx = rec[SSG_MAP_OFF_GOALS + 2 * g]; gy = rec[SSG_MAP_OFF_GOALS + 2 * g + 1]; }
            if ((x - gx) * (x - gx) + (y - gy) * (y - gy) <= c.goal_r * c.goal_r) { R = 0; G = 255; B = 0; }
        }
        if (in_ship(c.hull, c.nrm, px, py, ang, x, y)) { R = 255; G = 255; B = 255; }
        if (c.n_ships > 1)
            for (int k = 0; k < SSG_N_TRAFFIC; ++k) {
                const double *t = c.dyn_f64 + (size_t)(DC_TRAFFIC + 9 * k) * np + e;
                if (in_ship(d.thull[k], d.tnrm[k], t[0], t[np], t[2 * np], x, y)) { R = 0; G = 0; B = 0; }
            }
        const double sxr = c.width / width; // the circles have a radius of 10 screen pixels of the reference's screen
        for (int i = 0; i < c.n_beams; ++i) {
            const double ddx = x - beam_x[i], ddy = y - beam_y[i];
            if (ddx * ddx + ddy * ddy <= 100.0) { R = beam_hit[i] ? 255 : 0; G = beam_hit[i] ? 0 : 255; B = 0; }
        }
        (void)sxr;
    }
    if ((x - px) * (x - px) + (y - py) * (y - py) <= 100.0) { R = 255; G = 255; B = 0; }
    uint8_t *o = rgb + ((size_t)sx * height + sy) * 3;
    o[0] = R; o[1] = G; o[2] = B;
}

hipError_t launch_render(const DevCfg &c, const DynCfg &d, int e, int width, int height, uint8_t *rgb, unsigned flags,
                         hipStream_t stream)
{
    const int block = 128;
    hipLaunchKernelGGL(render_kernel, dim3((unsigned)((width + block - 1) / block), (unsigned)height), dim3(block), 0, stream, c, d,
                       e, width, height, rgb, flags);
    return hipGetLastError();
}

} // namespace ssg
