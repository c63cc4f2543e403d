/* rtb_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See rtb_oracle.h.
 *
 * A from-scratch C99 restatement of the arithmetic of the reference hot path; each routine
 * cites the reference file:line whose behaviour it follows.  Deliberately simple: row-major
 * 4x4s, full 4x4 products in the reference's association order, scalar loops, libm sin/cos.
 * Compiled with -ffp-contract=off so the flop sequence is the one written here.
 */
#include "rtb_oracle.h"
#include <math.h>
#include <string.h>

#define ORACLE_MAXN 64
static const double kPi = 3.14159265358979323846264338327950288; /* linalg.h:19 */
static const double kPi2 = 6.283185307179586;                    /* linalg.h:20 */
static const double kPiHalf = 1.57079632679489661923132169163975144; /* linalg.h:18 */

/* ---------------------------------------------------------------- small helpers */
static void m4_identity(double *M)
{
    for (int i = 0; i < 16; i++) M[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

static void m4_mul(const double *A, const double *B, double *C) /* C = A*B, C may not alias */
{
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += A[4 * r + k] * B[4 * k + c];
            C[4 * r + c] = s;
        }
}

/* Elementary transform of one chain entry.  kind 0..2: rotations (fknm.cpp:1320-1444),
 * 3..5: translations (fknm.cpp:1446-1555), 6: stored constant (methods.cpp:357-361). */
static void et_matrix(const oracle_chain *c, int i, const double *q, double *A)
{
    int k = c->kind[i];
    if (k == 6) {
        memcpy(A, c->consts + 16 * i, 16 * sizeof(double));
        return;
    }
    double eta = q[c->jindex[i]];
    if (c->flip[i]) eta = -eta; /* methods.cpp:363-366 */
    m4_identity(A);
    if (k <= 2) {
        double ct = cos(eta), st = sin(eta);
        int b = (k + 1) % 3, d = (k + 2) % 3; /* plane of rotation */
        A[4 * b + b] = ct;
        A[4 * b + d] = -st;
        A[4 * d + b] = st;
        A[4 * d + d] = ct;
    } else {
        A[4 * (k - 3) + 3] = eta;
    }
}

/* ---------------------------------------------------------------- forward kinematics */
void oracle_fkine(const oracle_chain *c, const double *q, long N, int q_stride,
                  const double *base16, const double *tool16, double *T)
{
    for (long s = 0; s < N; s++) {
        const double *qs = q + s * (long)q_stride;
        double cur[16], A[16], tmp[16];
        if (base16) memcpy(cur, base16, sizeof cur); else m4_identity(cur); /* methods.cpp:324-332 */
        for (int i = 0; i < c->m; i++) {                                    /* methods.cpp:334-341 */
            et_matrix(c, i, qs, A);
            m4_mul(cur, A, tmp);
            memcpy(cur, tmp, sizeof cur);
        }
        if (tool16) {                                                       /* methods.cpp:343-347 */
            m4_mul(cur, tool16, tmp);
            memcpy(cur, tmp, sizeof cur);
        }
        memcpy(T + 16 * s, cur, sizeof cur);
    }
}

/* ---------------------------------------------------------------- Jacobians */
static void jacob_one(const oracle_chain *c, const double *q, const double *tool16, int frame,
                      double *J /* 6*n row-major */)
{
    int n = c->n;
    double U[16], A[16], tmp[16];
    double Je[6 * ORACLE_MAXN];
    if (tool16) memcpy(U, tool16, sizeof U); else m4_identity(U); /* methods.cpp:116,123-128 */
    int j = n - 1;
    for (int i = c->m - 1; i >= 0; i--) {                          /* methods.cpp:130 */
        int k = c->kind[i];
        if (k != 6) {
            double col[6];
            if (k <= 2) {                                          /* methods.cpp:137-166 */
                int b = (k + 1) % 3, d = (k + 2) % 3;
                for (int x = 0; x < 3; x++) {
                    col[x] = U[4 * d + x] * U[4 * b + 3] - U[4 * b + x] * U[4 * d + 3];
                    col[3 + x] = U[4 * k + x];
                }
            } else {                                               /* methods.cpp:167-196 */
                for (int x = 0; x < 3; x++) {
                    col[x] = U[4 * (k - 3) + x];
                    col[3 + x] = 0.0;
                }
            }
            if (c->flip[i]) {
                int lim = (k <= 2) ? 6 : 3;
                for (int x = 0; x < lim; x++) col[x] = -col[x];
            }
            for (int x = 0; x < 6; x++) Je[x * n + j] = col[x];
            j--;
        }
        et_matrix(c, i, q, A);                                     /* methods.cpp:198-207 */
        m4_mul(A, U, tmp);
        memcpy(U, tmp, sizeof U);
    }
    if (frame == 1) {                                              /* jacobe: methods.cpp:219-316 */
        memcpy(J, Je, sizeof(double) * 6 * n);
        return;
    }
    for (int col = 0; col < n; col++)                              /* methods.cpp:211-216 */
        for (int half = 0; half < 2; half++)
            for (int r = 0; r < 3; r++) {
                double s = 0.0;
                for (int k = 0; k < 3; k++) s += U[4 * r + k] * Je[(3 * half + k) * n + col];
                J[(3 * half + r) * n + col] = s;
            }
}

void oracle_jacob(const oracle_chain *c, const double *q, long N, int q_stride,
                  const double *tool16, int frame, double *J)
{
    for (long s = 0; s < N; s++)
        jacob_one(c, q + s * (long)q_stride, tool16, frame, J + s * 6L * c->n);
}

static void cross3(const double *a, const double *b, double *r)
{
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

void oracle_hessian(int n, const double *J, double *H) /* methods.cpp:16-32 */
{
    memset(H, 0, sizeof(double) * n * 6 * n);
    for (int j = 0; j < n; j++)
        for (int i = j; i < n; i++) {
            double wj[3], vi[3], wi[3], a[3], b[3];
            for (int x = 0; x < 3; x++) {
                wj[x] = J[(3 + x) * n + j];
                vi[x] = J[x * n + i];
                wi[x] = J[(3 + x) * n + i];
            }
            cross3(wj, vi, a);
            cross3(wj, wi, b);
            for (int x = 0; x < 3; x++) {
                H[(j * 6 + x) * n + i] = a[x];
                H[(j * 6 + 3 + x) * n + i] = b[x];
                if (i != j) {
                    H[(i * 6 + x) * n + j] = a[x];
                    H[(i * 6 + 3 + x) * n + j] = 0.0;
                }
            }
        }
}

/* ---------------------------------------------------------------- pose error */
void oracle_angle_axis(const double *Te, const double *Tep, double *e) /* ik.cpp:241-286 */
{
    double R[9];
    for (int x = 0; x < 3; x++) e[x] = Tep[4 * x + 3] - Te[4 * x + 3];
    for (int r = 0; r < 3; r++) /* R = Rd * Re^T */
        for (int c = 0; c < 3; c++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s += Tep[4 * r + k] * Te[4 * c + k];
            R[3 * r + c] = s;
        }
    double li[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    double nrm = sqrt(li[0] * li[0] + li[1] * li[1] + li[2] * li[2]);
    double tr = R[0] + R[4] + R[8];
    if (nrm < 1e-6) {
        if (tr > 0) {
            e[3] = e[4] = e[5] = 0.0;
        } else {
            e[3] = kPiHalf * (R[0] + 1);
            e[4] = kPiHalf * (R[4] + 1);
            e[5] = kPiHalf * (R[8] + 1);
        }
    } else {
        double ang = atan2(nrm, tr - 1);
        for (int x = 0; x < 3; x++) e[3 + x] = ang * li[x] / nrm;
    }
}

/* ---------------------------------------------------------------- LM step shared by both IK flavours */
/* dq = (J^T We J + Wn)^-1 J^T We e with Wn = wn*I.  Dense solve by Gaussian elimination with
 * partial pivoting (the reference forms the explicit inverse with Eigen's partial-pivot LU,
 * ik.cpp:171,189,207; the two agree to rounding).  returns 0 if a zero pivot is met. */
static int lm_step(int n, const double *J, const double *e, const double *we, double wn, double *dq)
{
    double A[ORACLE_MAXN * (ORACLE_MAXN + 1)];
    int w = n + 1;
    for (int r = 0; r < n; r++) {
        for (int c = 0; c < n; c++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += J[k * n + r] * we[k] * J[k * n + c];
            A[r * w + c] = s + (r == c ? wn : 0.0);
        }
        double g = 0.0;
        for (int k = 0; k < 6; k++) g += J[k * n + r] * we[k] * e[k];
        A[r * w + n] = g;
    }
    for (int p = 0; p < n; p++) {
        int best = p;
        for (int r = p + 1; r < n; r++)
            if (fabs(A[r * w + p]) > fabs(A[best * w + p])) best = r;
        if (A[best * w + p] == 0.0) return 0;
        if (best != p)
            for (int c = 0; c < w; c++) {
                double t = A[p * w + c];
                A[p * w + c] = A[best * w + c];
                A[best * w + c] = t;
            }
        for (int r = p + 1; r < n; r++) {
            double f = A[r * w + p] / A[p * w + p];
            for (int c = p; c < w; c++) A[r * w + c] -= f * A[p * w + c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = A[r * w + n];
        for (int c = r + 1; c < n; c++) s -= A[r * w + c] * dq[c];
        dq[r] = s / A[r * w + r];
    }
    return 1;
}

static double lm_damping(int method, double lambda, double E)
{
    if (method == 1) return lambda;     /* Wampler  ik.cpp:183 */
    if (method == 2) return E + lambda; /* Sugihara ik.cpp:205 */
    return lambda * E;                  /* Chan     ik.cpp:169 */
}

/* q-vector scatter helpers: the chain reads q[jindex]; IK state is per joint ordinal. */
static void joint_columns(const oracle_chain *c, int *jcol)
{
    int j = 0;
    for (int i = 0; i < c->m; i++)
        if (c->kind[i] != 6) jcol[j++] = c->jindex[i];
}

void oracle_ik_lm(const oracle_chain *c, const double *qlim, const double *Tep, const double *q0,
                  int ilimit, int slimit, double tol, int reject_jl, const double *we6,
                  double lambda, int method, const double *restarts, double *q, int *solution,
                  int *it, int *search, double *E)
{
    /* The reference indexes q by jindex inside fkine/jacob0 but treats the IK state as a dense
     * n-vector (ik.cpp:34-37,57); for a chain with jindex == 0..n-1 (every serial arm) the two
     * coincide, which is the case restated here. */
    int n = c->n;
    double we[6], Te[16], e[6], J[6 * ORACLE_MAXN], dq[ORACLE_MAXN];
    for (int k = 0; k < 6; k++) we[k] = we6 ? we6[k] : 1.0;       /* ik.cpp:163 */
    int next = 0;
    *it = 0; *search = 1; *solution = 0; *E = 0.0;                /* fknm.cpp:406 */
    if (q0) memcpy(q, q0, sizeof(double) * n);                    /* ik.cpp:34-37 */
    else { memcpy(q, restarts + (long)n * next, sizeof(double) * n); next++; }
    int iter = 1;                                                 /* ik.cpp:39 */
    while (*search <= slimit) {
        while (iter <= ilimit) {
            oracle_fkine(c, q, 1, n, 0, 0, Te);
            oracle_angle_axis(Te, Tep, e);
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += e[k] * we[k] * e[k];
            *E = 0.5 * s;
            if (*E < tol) {                                        /* ik.cpp:48-54 */
                int ok = 1;
                for (int i = 0; i < n; i++) {
                    q[i] = fmod(q[i] + kPi, kPi2) - kPi;
                    if (q[i] < qlim[i] || q[i] > qlim[n + i]) ok = 0; /* ik.cpp:227-239 */
                }
                *solution = reject_jl ? ok : 1;
                break;
            }
            jacob_one(c, q, 0, 0, J);
            if (lm_step(n, J, e, we, lm_damping(method, lambda, *E), dq))
                for (int i = 0; i < n; i++) q[i] += dq[i];
            iter++;
        }
        if (*solution) { *it += iter; break; }                     /* ik.cpp:61-65 */
        *it += iter;
        iter = 0;
        (*search)++;
        memcpy(q, restarts + (long)n * next, sizeof(double) * n);  /* ik.cpp:69 */
        next++;
    }
}

static double py_mod(double a, double b) /* Python float % for b > 0 */
{
    double r = fmod(a, b);
    if (r < 0) r += b;
    return r;
}

void oracle_ikine_lm(const oracle_chain *c, const double *qlim, const double *Tep, const double *q0s,
                     int ilimit, int slimit, double tol, int joint_limits, const double *we6,
                     double k, int method, double *q_out, int *success, int *iterations,
                     int *searches, double *residual)
{
    int n = c->n;
    double we[6], Te[16], e[6], J[6 * ORACLE_MAXN], dq[ORACLE_MAXN], q[ORACLE_MAXN];
    for (int x = 0; x < 6; x++) we[x] = we6 ? we6[x] : 1.0;        /* IK.py:168-171 */
    int total = 0;
    double E = 0.0;
    memcpy(q, q0s, sizeof(double) * n);
    for (int s = 0; s < slimit; s++) {                              /* IK.py:311 */
        memcpy(q, q0s + (long)n * s, sizeof(double) * n);
        int i = 0;
        while (i < ilimit) {
            i++;
            oracle_fkine(c, q, 1, n, 0, 0, Te);                     /* IK.py:994-995 */
            oracle_angle_axis(Te, Tep, e);
            double acc = 0.0;
            for (int x = 0; x < 6; x++) acc += e[x] * we[x] * e[x];
            E = 0.5 * acc;
            jacob_one(c, q, 0, 0, J);
            if (lm_step(n, J, e, we, lm_damping(method, k, E), dq)) /* IK.py:997-1015 */
                for (int x = 0; x < n; x++) q[x] += dq[x];
            if (E < tol) {                                          /* IK.py:326-349 */
                int ok = 1;
                for (int x = 0; x < n; x++) {
                    q[x] = py_mod(q[x] + kPi, 2 * kPi) - kPi;
                    if (q[x] < qlim[x] || q[x] > qlim[n + x]) ok = 0;
                }
                if (!ok && joint_limits) break;
                memcpy(q_out, q, sizeof(double) * n);
                *success = 1; *iterations = total + i; *searches = s + 1; *residual = E;
                return;
            }
        }
        total += i;
    }
    memcpy(q_out, q, sizeof(double) * n);                           /* IK.py:359-366 */
    *success = 0; *iterations = total; *searches = slimit; *residual = E;
}

/* ---------------------------------------------------------------- Newton-Euler */
typedef struct { double x, y, z; } v3;
static v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static v3 v3_scale(v3 a, double s) { v3 r = {s * a.x, s * a.y, s * a.z}; return r; }
static v3 v3_cross(v3 a, v3 b)
{
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static double v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* R stored row-major r[3*row+col] */
static v3 rot_mul(const double *r, v3 v) /* R v      (vmath.c rot_vect_mult) */
{
    v3 o = {r[0] * v.x + r[1] * v.y + r[2] * v.z, r[3] * v.x + r[4] * v.y + r[5] * v.z,
            r[6] * v.x + r[7] * v.y + r[8] * v.z};
    return o;
}
static v3 rot_tmul(const double *r, v3 v) /* R^T v   (vmath.c rot_trans_vect_mult) */
{
    v3 o = {r[0] * v.x + r[3] * v.y + r[6] * v.z, r[1] * v.x + r[4] * v.y + r[7] * v.z,
            r[2] * v.x + r[5] * v.y + r[8] * v.z};
    return o;
}
static v3 inertia_mul(const double *I, v3 v) /* vmath.c mat_vect_mult, column-major m[r+3c] */
{
    v3 o = {I[0] * v.x + I[3] * v.y + I[6] * v.z, I[1] * v.x + I[4] * v.y + I[7] * v.z,
            I[2] * v.x + I[5] * v.y + I[8] * v.z};
    return o;
}

void oracle_rne_dh(const double *L, int n, int mdh, const double *q, const double *qd,
                   const double *qdd, long N, const double *grav3, const double *fext,
                   double *tau)
{
    const v3 zero = {0, 0, 0};
    v3 grav = {grav3[0], grav3[1], grav3[2]};
    for (long s = 0; s < N; s++) {
        const double *qs = q + s * n, *qds = qd + s * n, *qdds = qdd + s * n;
        double R[ORACLE_MAXN][9];
        v3 ps[ORACLE_MAXN], w[ORACLE_MAXN], wd[ORACLE_MAXN], a[ORACLE_MAXN], ac[ORACLE_MAXN];
        v3 f[ORACLE_MAXN], nn[ORACLE_MAXN];
        /* link rotation + offset vector: frne.c:193-207 and rot_mat frne.c:310-351 */
        for (int j = 0; j < n; j++) {
            const double *l = L + 24 * j;
            int prismatic = (int)l[4] != 0;
            double th = prismatic ? l[2] : qs[j] + l[5];
            double d = prismatic ? qs[j] + l[5] : l[3];
            double st = sin(th), ct = cos(th), sa = sin(l[0]), ca = cos(l[0]);
            double *r = R[j];
            if (!mdh) {
                r[0] = ct; r[1] = -ca * st; r[2] = sa * st;
                r[3] = st; r[4] = ca * ct;  r[5] = -sa * ct;
                r[6] = 0;  r[7] = sa;       r[8] = ca;
                ps[j].x = l[1]; ps[j].y = d * sa; ps[j].z = d * ca;
            } else {
                r[0] = ct;      r[1] = -st;     r[2] = 0;
                r[3] = st * ca; r[4] = ca * ct; r[5] = -sa;
                r[6] = st * sa; r[7] = ct * sa; r[8] = ca;
                ps[j].x = l[1]; ps[j].y = -d * sa; ps[j].z = d * ca;
            }
        }
        /* forward recursion: ne.c:133-348 */
        v3 qdv = zero, qddv = zero; /* only .z is rewritten per link (ne.c:140-141,248-249) */
        for (int j = 0; j < n; j++) {
            const double *l = L + 24 * j;
            int prismatic = (int)l[4] != 0;
            v3 rc = {l[7], l[8], l[9]};
            qdv.z = qds[j];
            qddv.z = qdds[j];
            if (mdh) {
                if (!prismatic) {
                    if (j == 0) {
                        w[j] = qdv; wd[j] = qddv;
                        a[j] = rot_tmul(R[j], grav);
                    } else {
                        v3 t1 = rot_tmul(R[j], w[j - 1]);
                        w[j] = v3_add(t1, qdv);
                        v3 t3 = rot_tmul(R[j], wd[j - 1]);
                        wd[j] = v3_add(v3_add(v3_cross(t1, qdv), t3), qddv);
                        v3 u = v3_cross(w[j - 1], v3_cross(w[j - 1], ps[j]));
                        v3 t = v3_add(v3_add(v3_cross(wd[j - 1], ps[j]), u), a[j - 1]);
                        a[j] = rot_tmul(R[j], t);
                    }
                } else {
                    if (j == 0) {
                        w[j] = qdv; wd[j] = qddv; a[j] = grav;
                    } else {
                        w[j] = rot_tmul(R[j], w[j - 1]);
                        wd[j] = rot_tmul(R[j], wd[j - 1]);
                        v3 u = v3_cross(w[j - 1], v3_cross(w[j - 1], ps[j]));
                        v3 t = v3_add(v3_add(v3_cross(wd[j - 1], ps[j]), u), a[j - 1]);
                        a[j] = rot_tmul(R[j], t);
                        v3 c2 = v3_scale(v3_cross(rot_tmul(R[j], w[j - 1]), qdv), 2.0);
                        a[j] = v3_add(v3_add(a[j], c2), qddv);
                    }
                }
            } else {
                if (!prismatic) {
                    v3 t1 = (j == 0) ? qdv : v3_add(w[j - 1], qdv);
                    w[j] = rot_tmul(R[j], t1);
                    v3 t3 = (j == 0) ? qddv
                                     : v3_add(v3_add(wd[j - 1], qddv), v3_cross(w[j - 1], qdv));
                    wd[j] = rot_tmul(R[j], t3);
                    v3 t = v3_add(v3_cross(wd[j], ps[j]), v3_cross(w[j], v3_cross(w[j], ps[j])));
                    a[j] = v3_add(t, rot_tmul(R[j], (j == 0) ? grav : a[j - 1]));
                } else {
                    w[j] = (j == 0) ? zero : rot_tmul(R[j], w[j - 1]);
                    wd[j] = (j == 0) ? zero : rot_tmul(R[j], wd[j - 1]);
                    if (j == 0) {
                        qddv = v3_add(qddv, grav); /* persists in x,y: ne.c:311 */
                        a[j] = rot_tmul(R[j], qddv);
                    } else {
                        a[j] = rot_tmul(R[j], v3_add(qddv, a[j - 1]));
                    }
                    a[j] = v3_add(a[j], v3_cross(wd[j], ps[j]));
                    v3 c2 = v3_scale(v3_cross(w[j], rot_tmul(R[j], qdv)), 2.0);
                    a[j] = v3_add(a[j], c2);
                    a[j] = v3_add(a[j], v3_cross(w[j], v3_cross(w[j], ps[j])));
                }
            }
            ac[j] = v3_add(v3_add(v3_cross(wd[j], rc), v3_cross(w[j], v3_cross(w[j], rc))), a[j]);
        }
        /* backward recursion: ne.c:354-458 */
        v3 ftip = zero, ntip = zero;
        if (fext) {
            ftip.x = fext[0]; ftip.y = fext[1]; ftip.z = fext[2];
            ntip.x = fext[3]; ntip.y = fext[4]; ntip.z = fext[5];
        }
        for (int j = n - 1; j >= 0; j--) {
            const double *l = L + 24 * j;
            v3 rc = {l[7], l[8], l[9]};
            v3 F = v3_scale(ac[j], l[6]);
            v3 Nn = v3_add(inertia_mul(l + 10, wd[j]), v3_cross(w[j], inertia_mul(l + 10, w[j])));
            int last = (j == n - 1);
            if (mdh) {
                v3 fn = last ? ftip : rot_mul(R[j + 1], f[j + 1]);
                f[j] = v3_add(fn, F);
                v3 t1 = last ? ntip
                             : v3_add(rot_mul(R[j + 1], nn[j + 1]), v3_cross(ps[j + 1], fn));
                nn[j] = v3_add(v3_add(t1, v3_cross(rc, F)), Nn);
            } else {
                f[j] = v3_add(F, last ? ftip : rot_mul(R[j + 1], f[j + 1]));
                v3 t1 = v3_cross(v3_add(ps[j], rc), F);
                if (!last) {
                    v3 t3 = v3_add(v3_cross(rot_tmul(R[j + 1], ps[j]), f[j + 1]), nn[j + 1]);
                    t1 = v3_add(t1, rot_mul(R[j + 1], t3));
                } else {
                    t1 = v3_add(v3_add(t1, v3_cross(ps[j], ftip)), ntip);
                }
                nn[j] = v3_add(t1, Nn);
            }
        }
        /* joint projection + actuator terms: ne.c:464-492 */
        for (int j = 0; j < n; j++) {
            const double *l = L + 24 * j;
            v3 z0 = {0, 0, 1};
            v3 ax = mdh ? z0 : rot_tmul(R[j], z0);
            double t = ((int)l[4] != 0) ? v3_dot(f[j], ax) : v3_dot(nn[j], ax);
            double G = l[20];
            t += G * G * l[19] * qdds[j];
            t += G * G * l[21] * qds[j];
            t += fabs(G) * ((qds[j] > 0 ? l[22] : 0.0) + (qds[j] < 0 ? l[23] : 0.0));
            tau[s * n + j] = t;
        }
    }
}

/* ---------------------------------------------------------------- DH closed form */
void oracle_dh_A(double alpha, double a, double theta, double d, int sigma, int mdh, double offset,
                 int flip, double q, double *T)
{
    double sa = sin(alpha), ca = cos(alpha);
    q = flip ? -q + offset : q + offset;                       /* DHLink.py:636-639 */
    double st, ct, dd;
    if (sigma == 0) { st = sin(q); ct = cos(q); dd = d; }      /* DHLink.py:641-650 */
    else { st = sin(theta); ct = cos(theta); dd = q; }
    m4_identity(T);
    if (!mdh) {                                                /* DHLink.py:652-661 */
        T[0] = ct; T[1] = -st * ca; T[2] = st * sa;  T[3] = a * ct;
        T[4] = st; T[5] = ct * ca;  T[6] = -ct * sa; T[7] = a * st;
        T[8] = 0;  T[9] = sa;       T[10] = ca;      T[11] = dd;
    } else {                                                   /* DHLink.py:662-671 */
        T[0] = ct;      T[1] = -st;     T[2] = 0;    T[3] = a;
        T[4] = st * ca; T[5] = ct * ca; T[6] = -sa;  T[7] = -sa * dd;
        T[8] = st * sa; T[9] = ct * sa; T[10] = ca;  T[11] = ca * dd;
    }
}

void oracle_dh_fkine(const double *dh, int n, int mdh, const double *q, long N,
                     const double *base16, const double *tool16, double *T)
{
    for (long s = 0; s < N; s++) {                             /* DHRobot.py:963-977 */
        double cur[16], A[16], tmp[16];
        for (int j = 0; j < n; j++) {
            const double *p = dh + 7 * j;
            oracle_dh_A(p[0], p[1], p[2], p[3], (int)p[4], mdh, p[5], (int)p[6], q[s * n + j], A);
            if (j == 0) memcpy(cur, A, sizeof cur);
            else { m4_mul(cur, A, tmp); memcpy(cur, tmp, sizeof cur); }
        }
        if (base16) { m4_mul(base16, cur, tmp); memcpy(cur, tmp, sizeof cur); }
        if (tool16) { m4_mul(cur, tool16, tmp); memcpy(cur, tmp, sizeof cur); }
        memcpy(T + 16 * s, cur, sizeof cur);
    }
}
