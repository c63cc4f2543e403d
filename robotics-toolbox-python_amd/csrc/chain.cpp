// chain.cpp -- host-side "chain compiler": lowers the elementary-transform list that the
// reference keeps as ET/ETS structs (core/structs.h:25-56, built by ET_init fknm.cpp:1182-1239
// and ETS_init fknm.cpp:1066-1114) to the canonical segment form the kernels execute
// (see rtbhip_internal.h).
//
// Design (MI355X-first, not a translation): the reference walks m ETs, building and multiplying a
// full 4x4 for each (methods.cpp:334-341) behind a function pointer per ET.  A GPU wants the
// opposite: no per-ET dispatch at all.  So (1) every run of constant ETs is folded into ONE affine
// (what the reference's optional ETS.compile() does, robot/ETS.py:857-906), and (2) every joint is
// rewritten to act on the local z axis: Rx(q) = M Rz(q) M^T and tx(q) = M tz(q) M^T for the cyclic
// axis permutation M that maps z to x (similarly y); M and M^T are absorbed into the constants on
// either side -- a column / row permutation, exact in floating point.  The device code is then the
// same straight-line sequence for every chain: n x { P <- P*C_j ; note axis and origin ;
// rotate two columns by q_j or slide along one }, then P <- P*C_n.
// Folding re-associates the constant products (rounding ~1e-16, against a 1e-10 parity budget).
#include "rtbhip_internal.h"
#include "frames_device.h"
#include <cmath>
#include <cstring>

namespace rtbhip {

namespace {

DevSeg seg_identity()
{
    DevSeg a;
    for (int i = 0; i < 9; i++) a.r[i] = (i % 4 == 0) ? 1.0 : 0.0;
    a.t[0] = a.t[1] = a.t[2] = 0.0;
    return a;
}

DevSeg seg_mul(const DevSeg &A, const DevSeg &B)
{
    DevSeg C;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s += A.r[3 * i + k] * B.r[3 * k + j];
            C.r[3 * i + j] = s;
        }
        double s = A.t[i];
        for (int k = 0; k < 3; k++) s += A.r[3 * i + k] * B.t[k];
        C.t[i] = s;
    }
    return C;
}

// cyclic axis permutation M_a with M_a e_z = e_a (a = 0 x, 1 y, 2 z); perm[c] = row holding the 1 of column c
void axis_perm(int a, int perm[3])
{
    // z -> a, x -> a+1, y -> a+2 (cyclic, det +1)
    perm[2] = a;
    perm[0] = (a + 1) % 3;
    perm[1] = (a + 2) % 3;
}

// A <- A * M  (columns of R permuted: new column c = old column perm[c]) -- exact
DevSeg right_perm(const DevSeg &A, const int perm[3])
{
    DevSeg O = A;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) O.r[3 * r + c] = A.r[3 * r + perm[c]];
    return O;
}

// returns M^T as a segment (to be left-multiplied into the next constant run) -- entries 0/1
DevSeg perm_transpose_seg(const int perm[3])
{
    DevSeg O;
    std::memset(&O, 0, sizeof O);
    // M[perm[c]][c] = 1  =>  M^T[c][perm[c]] = 1
    for (int c = 0; c < 3; c++) O.r[3 * c + perm[c]] = 1.0;
    return O;
}

// A FLIPPED joint (ET.py: eta = -q; methods.cpp:363-366 negates the coordinate, :142-145 / :172-175 the Jacobian column) is absorbed into the
// constants either side of it: with D = diag(1, -1, -1),  Z(-q) = D Z(q) D  for a rotation about z and for a slide along z alike (D turns the z
// axis round; sign changes are exact), so  C_j Z(-q) C_{j+1} = (C_j D) Z(q) (D C_{j+1}).  The device then walks an UNFLIPPED joint: its axis in
// the base frame is -z_j -- the negated Jacobian column the reference forms -- and every kernel's straight-line walk (and with it a structure
// signature, a run-time instantiation) serves robots with negative joint axes too (LBR iiwa joint 4, YuMi): no flip bit reaches the device.
DevSeg flip_columns(const DevSeg &A)      // A * D
{
    DevSeg O = A;
    for (int r = 0; r < 3; r++) { O.r[3 * r + 1] = -A.r[3 * r + 1]; O.r[3 * r + 2] = -A.r[3 * r + 2]; }
    return O;
}
// the constant run that follows a joint starts with M_a^T (the axis conjugation; the identity for a z joint), after D when the joint is flipped
void run_after_joint(int axis, bool flip, DevSeg *cur, bool *is_identity)
{
    int perm[3];
    axis_perm(axis, perm);
    if (axis == 2 && !flip) { *cur = seg_identity(); *is_identity = true; return; }
    DevSeg m = axis == 2 ? seg_identity() : perm_transpose_seg(perm);
    if (flip)
        for (int c = 0; c < 3; c++) { m.r[3 + c] = m.r[3 + c] == 0.0 ? 0.0 : -m.r[3 + c]; m.r[6 + c] = m.r[6 + c] == 0.0 ? 0.0 : -m.r[6 + c]; }   // D * m (zeros stay +0)
    *cur = m;
    *is_identity = false;
}

}  // namespace

// structure class + translation mask of a folded constant (rtbhip_internal.h: kSeg*), from its EXACT zeros and ones
int seg_class_bits(const DevSeg &a)
{
    const double *r = a.r;
    auto is = [&](int k, double v) { return r[k] == v; };
    int cls = kSegGeneral;
    if (is(0, 1) && is(1, 0) && is(2, 0) && is(3, 0) && is(4, 1) && is(5, 0) && is(6, 0) && is(7, 0) && is(8, 1)) cls = kSegIdentity;
    else if (is(0, 1) && is(1, 0) && is(2, 0) && is(3, 0) && is(6, 0))                 // row 0 and column 0 of an x rotation
        cls = (is(5, -1) && is(7, 1)) ? kSegRxP : (is(5, 1) && is(7, -1)) ? kSegRxN : kSegRx;
    else if (is(4, 1) && is(1, 0) && is(7, 0) && is(3, 0) && is(5, 0))                 // y
        cls = (is(2, -1) && is(6, 1)) ? kSegRyP : (is(2, 1) && is(6, -1)) ? kSegRyN : kSegRy;
    else if (is(8, 1) && is(2, 0) && is(5, 0) && is(6, 0) && is(7, 0))                 // z
        cls = (is(1, -1) && is(3, 1)) ? kSegRzP : (is(1, 1) && is(3, -1)) ? kSegRzN : kSegRz;
    else if (is(3, 1) && is(7, 1) && is(2, 1) && is(0, 0) && is(1, 0) && is(4, 0) && is(5, 0) && is(6, 0) && is(8, 0)) cls = kSegPermA;   // new columns = old (1, 2, 0)
    else if (is(6, 1) && is(1, 1) && is(5, 1) && is(0, 0) && is(2, 0) && is(3, 0) && is(4, 0) && is(7, 0) && is(8, 0)) cls = kSegPermB;   // new columns = old (2, 0, 1)
    const int tm = (a.t[0] != 0.0 ? 1 : 0) | (a.t[1] != 0.0 ? 2 : 0) | (a.t[2] != 0.0 ? 4 : 0);
    return (cls << 20) | (tm << 24);
}

int compile_chain(const rtbhip_et *ets, int m, const double *qlim, Chain *out)
{
    if (m < 0 || (m > 0 && ets == nullptr)) { set_error("chain_create: bad ets/m"); return RTBHIP_EINVAL; }
    if (m > RTBHIP_MAX_ETS) { set_error("chain_create: more than RTBHIP_MAX_ETS transforms"); return RTBHIP_ELIMIT; }
    out->ets.assign(ets, ets + m);
    out->seg.clear();
    out->jmeta.clear();
    DevSeg cur = seg_identity();  // constant run being accumulated
    bool cur_is_identity = true;
    int n = 0, qw = 0;
    std::vector<double> lo, hi;
    for (int i = 0; i < m; i++) {
        const rtbhip_et &e = ets[i];
        if (e.kind == RTBHIP_ET_CONST) {
            if (e.T[12] != 0.0 || e.T[13] != 0.0 || e.T[14] != 0.0 || e.T[15] != 1.0) {
                set_error("chain_create: constant transform " + std::to_string(i) +
                          " is not affine (bottom row must be 0 0 0 1)");
                return RTBHIP_EINVAL;
            }
            DevSeg a;
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) a.r[3 * r + c] = e.T[4 * r + c];
                a.t[r] = e.T[4 * r + 3];
            }
            cur = cur_is_identity ? a : seg_mul(cur, a);
            cur_is_identity = false;
            continue;
        }
        if (e.kind < 0 || e.kind > 5) {
            set_error("chain_create: unknown transform kind " + std::to_string(e.kind));
            return RTBHIP_EINVAL;
        }
        if (e.jindex < 0 || e.jindex > 255) {
            set_error("chain_create: jindex out of range (0..255)");
            return RTBHIP_EINVAL;
        }
        if (n == RTBHIP_MAX_JOINTS) { set_error("chain_create: more than RTBHIP_MAX_JOINTS joints"); return RTBHIP_ELIMIT; }
        const int axis = e.kind % 3;
        const bool prismatic = e.kind >= 3;
        int perm[3];
        axis_perm(axis, perm);
        const DevSeg cj = axis == 2 ? cur : right_perm(cur, perm);      // C_j * M_a
        out->seg.push_back(e.flip ? flip_columns(cj) : cj);             // ... * D for a flipped joint
        out->jmeta.push_back((prismatic ? 1 : 0) | (e.jindex << 8));    // (no flip bit: absorbed above)
        run_after_joint(axis, e.flip != 0, &cur, &cur_is_identity);     // (D) M_a^T starts the next run
        n++;
        if (e.jindex + 1 > qw) qw = e.jindex + 1;
        lo.push_back(prismatic ? 0.0 : -M_PI);  // default limits: robot/ET.py:109-115
        hi.push_back(prismatic ? 1.0 : M_PI);
    }
    out->seg.push_back(cur);  // C_n (identity when the chain ends with a z joint)
    for (int j = 0; j < n; j++) out->jmeta[j] |= seg_class_bits(out->seg[j]);
    out->jmeta.push_back(seg_class_bits(out->seg[n]));      // descriptor n: the tail's class alone
    out->n = n;
    out->q_width = qw;
    out->qlim.resize(2 * (size_t)n);
    for (int j = 0; j < n; j++) {
        out->qlim[j] = qlim ? qlim[j] : lo[j];
        out->qlim[n + j] = qlim ? qlim[n + j] : hi[j];
    }
    return RTBHIP_OK;
}

// Product-of-exponentials chains (reference robot/PoERobot.py): T(q) = exp([S_1] q_1) ... exp([S_n] q_n) T0 with unit joint
// twists S_i = (v_i, w_i) given in the BASE frame (PoERevolute: w = unit axis, v = -w x point, Twist3.UnitRevolute;
// PoEPrismatic: w = 0, v = unit direction).  The reference evaluates that product with one matrix exponential per joint in
// Python (PoERobot.fkine :209-228, jacob0 :230-250, jacobe :252-270) and, for everything else, re-expresses the robot as an
// ETS through roll-pitch-yaw angles (_update_ets :272-324).  Here the twists are lowered DIRECTLY to the canonical segment
// form: with W_i any frame whose z axis is the screw axis and whose origin lies on it, exp([S_i] q) = W_i Z(q) W_i^-1
// (Z = rotation about / translation along z), so the product telescopes to
//        T(q) = (W_1) Z(q_1) (W_1^-1 W_2) Z(q_2) ... Z(q_n) (W_n^-1 T0)
// -- n joints about z and n + 1 constants, no exponential, no angle extraction, nothing the device code does not already run.
int compile_poe(const double *twists, int n, const double *T0, const double *qlim, Chain *out)
{
    if (n < 0 || (n > 0 && twists == nullptr)) { set_error("chain_create_poe: bad twists/n"); return RTBHIP_EINVAL; }
    if (n > RTBHIP_MAX_JOINTS) { set_error("chain_create_poe: more than RTBHIP_MAX_JOINTS joints"); return RTBHIP_ELIMIT; }
    if (T0 && (T0[12] != 0.0 || T0[13] != 0.0 || T0[14] != 0.0 || T0[15] != 1.0)) {
        set_error("chain_create_poe: T0 is not affine (bottom row must be 0 0 0 1)");
        return RTBHIP_EINVAL;
    }
    std::vector<rtbhip_et> ets;
    ets.reserve(2 * (size_t)n + 1);
    double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, cp[3] = {0, 0, 0};   // W_{i-1}: rotation (row-major), origin
    auto push_const = [&](const double R[9], const double c[3]) {      // W_{i-1}^-1 * [R | c]
        rtbhip_et e;
        std::memset(&e, 0, sizeof e);
        e.kind = RTBHIP_ET_CONST;
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) {
                double s = 0.0;
                for (int j = 0; j < 3; j++) s += Rp[3 * j + r] * R[3 * j + k];
                e.T[4 * r + k] = s;
            }
            double s = 0.0;
            for (int j = 0; j < 3; j++) s += Rp[3 * j + r] * (c[j] - cp[j]);
            e.T[4 * r + 3] = s;
        }
        e.T[15] = 1.0;
        ets.push_back(e);
    };
    for (int i = 0; i < n; i++) {
        const double *v = twists + 6 * (size_t)i, *w = v + 3;
        const double wn = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double vn = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        double a[3], c[3] = {0, 0, 0};
        bool prismatic;
        if (!(wn == wn) || !(vn == vn)) { set_error("chain_create_poe: twist " + std::to_string(i) + " is not finite"); return RTBHIP_EINVAL; }
        if (wn == 0.0) {
            if (std::fabs(vn - 1.0) > 1e-9) { set_error("chain_create_poe: twist " + std::to_string(i) + " is not a unit prismatic twist (w = 0 needs |v| = 1)"); return RTBHIP_EINVAL; }
            prismatic = true;
            for (int k = 0; k < 3; k++) a[k] = v[k] / vn;
        } else {
            if (std::fabs(wn - 1.0) > 1e-9) { set_error("chain_create_poe: twist " + std::to_string(i) + " is not a unit revolute twist (|w| must be 1)"); return RTBHIP_EINVAL; }
            const double pitch = (w[0] * v[0] + w[1] * v[1] + w[2] * v[2]) / (wn * wn);
            if (std::fabs(pitch) > 1e-9 * (1.0 + vn)) { set_error("chain_create_poe: twist " + std::to_string(i) + " has a pitch (w . v != 0): only pure revolute / prismatic joints"); return RTBHIP_EINVAL; }
            prismatic = false;
            for (int k = 0; k < 3; k++) a[k] = w[k] / wn;
            // the point of the axis nearest the origin: w x v / |w|^2  (PoERobot.py:75 `principal_point`)
            c[0] = (w[1] * v[2] - w[2] * v[1]) / (wn * wn);
            c[1] = (w[2] * v[0] - w[0] * v[2]) / (wn * wn);
            c[2] = (w[0] * v[1] - w[1] * v[0]) / (wn * wn);
        }
        // x: the coordinate axis least aligned with a, made orthogonal to it; y = a x x
        int k0 = 0;
        if (std::fabs(a[1]) < std::fabs(a[k0])) k0 = 1;
        if (std::fabs(a[2]) < std::fabs(a[k0])) k0 = 2;
        double x[3] = {0, 0, 0};
        x[k0] = 1.0;
        const double d = a[k0];
        double xn = 0.0;
        for (int k = 0; k < 3; k++) { x[k] -= d * a[k]; xn += x[k] * x[k]; }
        xn = std::sqrt(xn);
        for (int k = 0; k < 3; k++) x[k] /= xn;
        const double y[3] = {a[1] * x[2] - a[2] * x[1], a[2] * x[0] - a[0] * x[2], a[0] * x[1] - a[1] * x[0]};
        double R[9];
        for (int r = 0; r < 3; r++) { R[3 * r] = x[r]; R[3 * r + 1] = y[r]; R[3 * r + 2] = a[r]; }
        push_const(R, c);
        rtbhip_et j;
        std::memset(&j, 0, sizeof j);
        j.kind = prismatic ? RTBHIP_ET_TZ : RTBHIP_ET_RZ;
        j.jindex = i;
        for (int k = 0; k < 4; k++) j.T[5 * k] = 1.0;
        ets.push_back(j);
        std::memcpy(Rp, R, sizeof Rp);
        std::memcpy(cp, c, sizeof cp);
    }
    {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, c[3] = {0, 0, 0};
        if (T0)
            for (int r = 0; r < 3; r++) {
                for (int k = 0; k < 3; k++) R[3 * r + k] = T0[4 * r + k];
                c[r] = T0[4 * r + 3];
            }
        push_const(R, c);
    }
    return compile_chain(ets.data(), (int)ets.size(), qlim, out);
}

// marks[m] = k: frame m is the product of the first k transforms of the chain.  Re-runs the folding of compile_chain and
// records, at each mark, how many joints precede it and the constant run accumulated since the last joint.
int compile_frames(const Chain *c, const int32_t *marks, int nmarks, FrameTable *ft)
{
    if (nmarks < 0 || nmarks > kMaxFrames) { set_error("link_frames: at most " + std::to_string(kMaxFrames) + " frames per call"); return RTBHIP_ELIMIT; }
    const int m = (int)c->ets.size();
    ft->nmarks = nmarks;
    int next = 0;
    for (int k = 0; k < nmarks; ++k) {
        if (marks[k] < 0 || marks[k] > m || (k > 0 && marks[k] < marks[k - 1])) {
            set_error("link_frames: marks must be nondecreasing transform counts in 0..m");
            return RTBHIP_EINVAL;
        }
    }
    DevSeg cur = seg_identity();
    bool cur_is_identity = true;
    int n = 0;
    auto record = [&](int done) {
        while (next < nmarks && marks[next] == done) {
            ft->jcount[next] = n;
            ft->ident[next] = cur_is_identity ? 1 : 0;
            for (int i = 0; i < 9; i++) ft->F[next][i] = cur.r[i];
            for (int i = 0; i < 3; i++) ft->F[next][9 + i] = cur.t[i];
            ++next;
        }
    };
    record(0);
    for (int i = 0; i < m; i++) {
        const rtbhip_et &e = c->ets[i];
        if (e.kind == RTBHIP_ET_CONST) {
            DevSeg a;
            for (int r = 0; r < 3; r++) {
                for (int k = 0; k < 3; k++) a.r[3 * r + k] = e.T[4 * r + k];
                a.t[r] = e.T[4 * r + 3];
            }
            cur = cur_is_identity ? a : seg_mul(cur, a);
            cur_is_identity = false;
        } else {
            run_after_joint(e.kind % 3, e.flip != 0, &cur, &cur_is_identity);
            n++;
        }
        record(i + 1);
    }
    return RTBHIP_OK;
}

// tail = C_n * tool (per-call tool folded on the host; tool may be unused => C_n)
void chain_tail(const Chain *c, const Affine &tool, double out12[12])
{
    DevSeg t;
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) t.r[3 * r + k] = tool.v[4 * r + k];
        t.t[r] = tool.v[4 * r + 3];
    }
    DevSeg res = tool.used ? seg_mul(c->seg.back(), t) : c->seg.back();
    for (int i = 0; i < 9; i++) out12[i] = res.r[i];
    for (int i = 0; i < 3; i++) out12[9 + i] = res.t[i];
}

}  // namespace rtbhip
