// frames_device.h -- poses of intermediate frames of a chain ("fkine_all": DHRobot.fkine_all robot/DHRobot.py:1012-1064,
// Robot.fkine_all robot/Robot.py:638-698).  The reference multiplies link matrices one by one in Python and keeps every
// partial product.  Here a frame is a MARK in the elementary-transform list: "the product of the first k transforms".
// In the canonical segment form (chain.cpp) that is  P_j * F  with P_j the walk's pose right after joint j-1 (j = number
// of joints among the first k transforms) and F the constant run accumulated since -- including the axis permutation a
// conjugated x/y joint left behind.  The host lowers the marks to (j, F) pairs; one lane walks the chain once and emits
// every frame on the way.
#pragma once
#include "kin_device.h"

namespace rtbhip {

constexpr int kMaxFrames = 33;                 // base + one frame per joint of the longest chain

struct FrameTable {                            // kernarg, wave-uniform
    int32_t nmarks, has_base;
    int32_t jcount[kMaxFrames + 1];            // joints before mark m (nondecreasing)
    int32_t ident[kMaxFrames + 1];             // F is the identity: emit the walk's pose as it is
    double F[kMaxFrames][12];                  // {R row-major (9), t (3)}
    double base[12];                           // row-major 3x4
};

// emit(m, pose) is called once per mark, in mark order
template <class CV, class QCol, class Emit>
RTB_HD void frames_walk(const CV &cv, int n, const FrameTable &ft, QCol qcol, Emit emit)
{
    Pose P;
    pose_identity(P);
    if (ft.has_base) {
        P.r00 = ft.base[0]; P.r01 = ft.base[1]; P.r02 = ft.base[2]; P.tx = ft.base[3];
        P.r10 = ft.base[4]; P.r11 = ft.base[5]; P.r12 = ft.base[6]; P.ty = ft.base[7];
        P.r20 = ft.base[8]; P.r21 = ft.base[9]; P.r22 = ft.base[10]; P.tz = ft.base[11];
    }
    int m = 0;
    auto drain = [&](int jc) {
        while (m < ft.nmarks && ft.jcount[m] == jc) {
            Pose Q = P;
            if (!ft.ident[m]) pose_mul_general(Q, [&](int k) { return ft.F[m][k]; });
            emit(m, Q);
            ++m;
        }
    };
    drain(0);
    for (int j = 0; j < n; ++j) {
        pose_mul_seg(P, cv, j);
        const int jm = cv.jmeta[j];
        double eta = qcol(jm_jq(jm));
        if (jm_flip(jm)) eta = -eta;
        if (jm_prismatic(jm)) {
            pose_tz(P, eta);
        } else {
            double s, c;
            rtb_sincos(eta, &s, &c);
            pose_rotz(P, c, s);
        }
        drain(j + 1);
    }
}

}  // namespace rtbhip
