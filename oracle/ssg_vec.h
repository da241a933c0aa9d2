/*
 * ssg_vec.h — cpVect / cpTransform helpers shared by the oracle's translation units.
 * TEST INFRASTRUCTURE ONLY (see ssg_oracle.h).
 */
#ifndef SSG_VEC_H
#define SSG_VEC_H
#include <float.h>
#include <math.h>
#include "ssg_oracle.h"

/* ------------------------------------------------------------------------------------------------
 * cpVect helpers (chipmunk/cpVect.h)
 * ---------------------------------------------------------------------------------------------- */
static inline ora_v2 V(double x, double y) { ora_v2 r = {x, y}; return r; }
static inline ora_v2 vadd(ora_v2 a, ora_v2 b) { return V(a.x + b.x, a.y + b.y); }
static inline ora_v2 vsub(ora_v2 a, ora_v2 b) { return V(a.x - b.x, a.y - b.y); }
static inline ora_v2 vmult(ora_v2 a, double s) { return V(a.x * s, a.y * s); }
static inline double vdot(ora_v2 a, ora_v2 b) { return a.x * b.x + a.y * b.y; }
static inline double vcross(ora_v2 a, ora_v2 b) { return a.x * b.y - a.y * b.x; }
static inline ora_v2 vrperp(ora_v2 a) { return V(a.y, -a.x); }
static inline double vlength(ora_v2 a) { return sqrt(vdot(a, a)); }
static inline ora_v2 vlerp(ora_v2 a, ora_v2 b, double t) { return vadd(vmult(a, 1.0 - t), vmult(b, t)); }
static inline ora_v2 vnormalize(ora_v2 a) { return vmult(a, 1.0 / (vlength(a) + DBL_MIN)); }
static inline double vdist(ora_v2 a, ora_v2 b) { return vlength(vsub(a, b)); }
static inline double fclamp01(double f) { return fmax(0.0, fmin(f, 1.0)); }

/* cpTransformPoint / cpTransformVect for the rigid transform built by cpBody SetTransform with cog=(0,0):
 *   a = rot.x, b = rot.y, c = -rot.y, d = rot.x, tx = p.x, ty = p.y                                 */
static inline ora_v2 xf_point(ora_v2 p, ora_v2 rot, ora_v2 v)
{
    return V(rot.x * v.x + (-rot.y) * v.y + p.x, rot.y * v.x + rot.x * v.y + p.y);
}
static inline ora_v2 xf_vect(ora_v2 rot, ora_v2 v)
{
    return V(rot.x * v.x + (-rot.y) * v.y, rot.y * v.x + rot.x * v.y);
}
static inline ora_v2 vperp(ora_v2 a) { return V(-a.y, a.x); }
static inline ora_v2 vneg(ora_v2 a) { return V(-a.x, -a.y); }
static inline double vlengthsq(ora_v2 a) { return vdot(a, a); }
static inline ora_v2 vrotate(ora_v2 a, ora_v2 b) { return V(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
/* cpfmin / cpfmax / cpfclamp exactly as chipmunk_types.h spells them (ternaries, not fmin/fmax) */
static inline double cfmin(double a, double b) { return (a < b) ? a : b; }
static inline double cfmax(double a, double b) { return (a > b) ? a : b; }
static inline double cfclamp(double f, double lo, double hi) { return cfmin(cfmax(f, lo), hi); }
static inline double cfclamp01(double f) { return cfmax(0.0, cfmin(f, 1.0)); }

#endif
