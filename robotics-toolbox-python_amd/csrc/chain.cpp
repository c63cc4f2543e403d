// chain.cpp -- host-side "chain compiler": lowers the elementary-transform list that the
// reference keeps as ET/ETS structs (core/structs.h:25-56, built by ET_init fknm.cpp:1182-1239
// and ETS_init fknm.cpp:1066-1114) to the flat device program the kernels interpret.
//
// Design (MI355X-first, not a translation): the reference multiplies a full 4x4 for every ET
// (methods.cpp:334-341).  Here each ET is classified once, on the host, into the cheapest update of
// a 3x4 affine pose held in registers: a constant rotation about one axis touches two columns
// (12 flops), a single-axis translation one column (3 FMAs), a general constant the full 36.  Runs of
// adjacent constants are folded into one op when -- and only when -- the folded op is cheaper than
// the specialised sequence (the reference's ETS.compile(), robot/ETS.py:857-906, always folds).
#include "rtbhip_internal.h"
#include <cmath>
#include <cstring>

namespace rtbhip {

namespace {

struct Aff {  // row-major 3x4
    double r[9];
    double t[3];
};

Aff aff_identity()
{
    Aff a;
    for (int i = 0; i < 9; i++) a.r[i] = (i % 4 == 0) ? 1.0 : 0.0;
    a.t[0] = a.t[1] = a.t[2] = 0.0;
    return a;
}

Aff aff_mul(const Aff &A, const Aff &B)
{
    Aff C;
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

bool rot_is_identity(const Aff &a)
{
    for (int i = 0; i < 9; i++)
        if (a.r[i] != ((i % 4 == 0) ? 1.0 : 0.0)) return false;
    return true;
}

// exact structural test for a rotation about coordinate axis k (what trotx/y/z produce)
bool rot_about_axis(const Aff &a, int k, double *c, double *s)
{
    int b = (k + 1) % 3, d = (k + 2) % 3;
    const double *r = a.r;
    if (r[3 * k + k] != 1.0) return false;
    if (r[3 * k + b] != 0.0 || r[3 * k + d] != 0.0 || r[3 * b + k] != 0.0 || r[3 * d + k] != 0.0)
        return false;
    if (r[3 * b + b] != r[3 * d + d]) return false;
    if (r[3 * b + d] != -r[3 * d + b]) return false;
    *c = r[3 * b + b];
    *s = r[3 * d + b];
    return true;
}

// Lower one constant affine to 0..2 device ops; returns the flop cost it adds.
int emit_const(const Aff &a, std::vector<DevOp> *out)
{
    DevOp op;
    std::memset(&op, 0, sizeof op);
    bool t0 = (a.t[0] == 0.0 && a.t[1] == 0.0 && a.t[2] == 0.0);
    if (rot_is_identity(a)) {
        int nz = (a.t[0] != 0.0) + (a.t[1] != 0.0) + (a.t[2] != 0.0);
        if (nz == 0) return 0;  // identity: nothing to do
        if (nz == 1) {
            int k = (a.t[0] != 0.0) ? 0 : (a.t[1] != 0.0) ? 1 : 2;
            op.kind = K_CTX + k;
            op.p[0] = a.t[k];
            if (out) out->push_back(op);
            return 3;
        }
        op.kind = K_CT3;
        op.p[0] = a.t[0]; op.p[1] = a.t[1]; op.p[2] = a.t[2];
        if (out) out->push_back(op);
        return 9;
    }
    if (t0) {
        for (int k = 0; k < 3; k++) {
            double c, s;
            if (rot_about_axis(a, k, &c, &s)) {
                op.kind = K_CRX + k;
                op.p[0] = c; op.p[1] = s;
                if (out) out->push_back(op);
                return 12;
            }
        }
    }
    op.kind = K_CGEN;
    for (int i = 0; i < 9; i++) op.p[i] = a.r[i];
    for (int i = 0; i < 3; i++) op.p[9 + i] = a.t[i];
    if (out) out->push_back(op);
    return 36;
}

void flush_run(std::vector<Aff> *run, std::vector<DevOp> *ops)
{
    if (run->empty()) return;
    int separate = 0;
    for (const Aff &a : *run) separate += emit_const(a, nullptr) + 2;  // +2: interpreter dispatch
    Aff folded = (*run)[0];
    for (size_t i = 1; i < run->size(); i++) folded = aff_mul(folded, (*run)[i]);
    int together = emit_const(folded, nullptr) + 2;
    if (together < separate) emit_const(folded, ops);
    else for (const Aff &a : *run) emit_const(a, ops);
    run->clear();
}

}  // namespace

int compile_chain(const rtbhip_et *ets, int m, const double *qlim, Chain *out)
{
    if (m < 0 || (m > 0 && ets == nullptr)) { set_error("chain_create: bad ets/m"); return RTBHIP_EINVAL; }
    if (m > RTBHIP_MAX_ETS) { set_error("chain_create: more than RTBHIP_MAX_ETS transforms"); return RTBHIP_ELIMIT; }
    out->ets.assign(ets, ets + m);
    out->ops.clear();
    std::vector<Aff> run;
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
            Aff a;
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) a.r[3 * r + c] = e.T[4 * r + c];
                a.t[r] = e.T[4 * r + 3];
            }
            run.push_back(a);
            continue;
        }
        if (e.kind < 0 || e.kind > 5) {
            set_error("chain_create: unknown transform kind " + std::to_string(e.kind));
            return RTBHIP_EINVAL;
        }
        if (e.jindex < 0 || e.jindex >= 4 * RTBHIP_MAX_JOINTS) {
            set_error("chain_create: jindex out of range");
            return RTBHIP_EINVAL;
        }
        flush_run(&run, &out->ops);
        DevOp op;
        std::memset(&op, 0, sizeof op);
        op.kind = e.kind;  // K_JRX.. == RTBHIP_ET_RX..
        op.jq = e.jindex;
        op.jcol = n++;
        op.flip = e.flip ? 1 : 0;
        out->ops.push_back(op);
        if (e.jindex + 1 > qw) qw = e.jindex + 1;
        bool rot = e.kind <= 2;  // default limits: robot/ET.py:109-115
        lo.push_back(rot ? -M_PI : 0.0);
        hi.push_back(rot ? M_PI : 1.0);
    }
    flush_run(&run, &out->ops);
    if (n > RTBHIP_MAX_JOINTS) { set_error("chain_create: more than RTBHIP_MAX_JOINTS joints"); return RTBHIP_ELIMIT; }
    out->n = n;
    out->q_width = qw;
    out->qlim.resize(2 * (size_t)n);
    for (int j = 0; j < n; j++) {
        out->qlim[j] = qlim ? qlim[j] : lo[j];
        out->qlim[n + j] = qlim ? qlim[n + j] : hi[j];
    }
    return RTBHIP_OK;
}

}  // namespace rtbhip
