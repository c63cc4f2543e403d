// tree.cpp -- host-side compiler of an ETS-robot dynamics tree (link groups of Robot.rne, reference
// robot/Robot.py:1777-1800) into the device table of tree_device.h: joints conjugated to the local z
// axis (the axis permutation is absorbed into the group's constant, its inertia and its children's
// constants, exactly as chain.cpp does for kinematic chains) and LDS slots assigned to the groups
// whose state is needed by a non-adjacent child.
#include "tree_device.h"
#include <cstring>

namespace rtbhip {

namespace {
void perm_of_axis(int a, int perm[3]) { perm[2] = a; perm[0] = (a + 1) % 3; perm[1] = (a + 2) % 3; }   // M e_z = e_a
}  // namespace

// structure signature of a compiled tree (tree_device.h): class and translation mask of every group constant from its EXACT zeros and ones
// (chain.cpp: seg_class_bits), plus kTreeSigPlain for a serial chain of revolute joints numbered in group order
// (trees of 9 .. 16 groups: the fields of groups 8 .. 15 go to a second word, *sig2; such a tree is never "plain")
SegSig tree_signature(const DevGroup *g, int ng, SegSig *sig2)
{
    if (sig2) *sig2 = 0;
    if (ng < 1 || ng > (sig2 ? kTreeSig2MaxGroups : kTreeSigMaxGroups)) return 0;
    SegSig s = kSegSigPresent;
    bool plain = ng <= kTreeSigMaxGroups;
    for (int j = 0; j < ng; j++) {
        const int bits = seg_class_bits(g[j].C);
        if (j >= kTreeSigMaxGroups) { *sig2 |= kSegSigPresent | seg_sig_of(j - kTreeSigMaxGroups, jm_cls(bits), jm_tmask(bits)); continue; }
        s |= seg_sig_of(j, jm_cls(bits), jm_tmask(bits));
        plain = plain && g[j].parent == j - 1 && g[j].save_slot < 0 && g[j].parent_slot < 0 && !jm_prismatic(g[j].jmeta) && jm_jq(g[j].jmeta) == j &&
                g[j].out_col == j;
    }
    return plain ? (s | kTreeSigPlain) : s;
}

// the bookkeeping of a compiled tree as one word (tree_device.h: TreeTopo): trees of up to 10 groups numbered in group order, at most 6 branch slots
TreeTopo tree_topology(const DevGroup *g, int ng, int nslots)
{
    if (ng < 1 || ng > kTreeTopoMaxGroups || nslots > 6) return 0;
    TreeTopo t = kTreeTopoPresent;
    for (int j = 0; j < ng; j++) {
        if (jm_jq(g[j].jmeta) != j || g[j].out_col != j || g[j].parent < -1 || g[j].parent >= j) return 0;
        t |= topo_of(j, g[j].parent, jm_prismatic(g[j].jmeta) != 0, g[j].parent_slot, g[j].save_slot);
    }
    return t;
}

int compile_tree(const rtbhip_tree_group *in, int ng, Tree *out)
{
    if (ng < 1 || in == nullptr) { set_error("tree_create: need at least one group"); return RTBHIP_EINVAL; }
    if (ng > RTBHIP_MAX_JOINTS) { set_error("tree_create: more than RTBHIP_MAX_JOINTS groups"); return RTBHIP_ELIMIT; }
    out->groups.assign(ng, DevGroup());
    std::vector<int> seen(ng, 0);
    // per-group axis permutation to apply to the children's constants
    std::vector<int> axis(ng, 2);
    for (int j = 0; j < ng; j++) {
        const rtbhip_tree_group &g = in[j];
        if (g.parent >= j || g.parent < -1) { set_error("tree_create: groups must be in topological order (parent < child)"); return RTBHIP_EINVAL; }
        if (g.kind < 0 || g.kind > 5) { set_error("tree_create: joint kind must be 0..5 (Rx,Ry,Rz,tx,ty,tz)"); return RTBHIP_EINVAL; }
        if (g.jindex < 0 || g.jindex >= ng || seen[g.jindex]) { set_error("tree_create: jindex must be a permutation of 0..ng-1"); return RTBHIP_EINVAL; }
        if (g.T[12] != 0.0 || g.T[13] != 0.0 || g.T[14] != 0.0 || g.T[15] != 1.0) { set_error("tree_create: group transform is not affine"); return RTBHIP_EINVAL; }
        seen[g.jindex] = 1;
        axis[j] = g.kind % 3;
        DevGroup &d = out->groups[j];
        double R[9], t[3];
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = g.T[4 * r + c]; t[r] = g.T[4 * r + 3]; }
        // left-multiply by the parent's M^T (row permutation) when the parent joint was not about z
        if (g.parent >= 0 && axis[g.parent] != 2) {
            int perm[3];
            perm_of_axis(axis[g.parent], perm);
            double R2[9], t2[3];
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R2[3 * r + c] = R[3 * perm[r] + c]; t2[r] = t[perm[r]]; }   // (M^T X)[r] = X[perm[r]]
            std::memcpy(R, R2, sizeof R); std::memcpy(t, t2, sizeof t);
        }
        double h[3] = {g.h[0], g.h[1], g.h[2]};
        double I[3][3] = {{g.I[0], g.I[3], g.I[4]}, {g.I[3], g.I[1], g.I[5]}, {g.I[4], g.I[5], g.I[2]}};
        if (axis[j] != 2) {   // own joint: C <- C M (column permutation), inertia re-expressed in the permuted frame
            int perm[3];
            perm_of_axis(axis[j], perm);
            double R2[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R2[3 * r + c] = R[3 * r + perm[c]];
            std::memcpy(R, R2, sizeof R);
            double h2[3], I2[3][3];
            for (int r = 0; r < 3; r++) { h2[r] = h[perm[r]]; for (int c = 0; c < 3; c++) I2[r][c] = I[perm[r]][perm[c]]; }
            std::memcpy(h, h2, sizeof h); std::memcpy(I, I2, sizeof I);
        }
        for (int k = 0; k < 9; k++) d.C.r[k] = R[k];
        for (int k = 0; k < 3; k++) { d.C.t[k] = t[k]; d.h[k] = h[k]; }
        d.M = g.m;
        d.I[0] = I[0][0]; d.I[1] = I[1][1]; d.I[2] = I[2][2]; d.I[3] = I[0][1]; d.I[4] = I[0][2]; d.I[5] = I[1][2];
        d.parent = g.parent;
        d.jmeta = (g.kind >= 3 ? 1 : 0) | (g.jindex << 8) | ((g.flip ? 1 : 0) << 16);
        d.save_slot = -1; d.parent_slot = -1; d.out_col = j;
        d.pad[0] = d.pad[1] = d.pad[2] = 0;
    }
    int nslots = 0;
    for (int j = 0; j < ng; j++) {
        const int p = out->groups[j].parent;
        if (p >= 0 && p != j - 1) {
            if (out->groups[p].save_slot < 0) out->groups[p].save_slot = nslots++;
            out->groups[j].parent_slot = out->groups[p].save_slot;
        }
    }
    out->n = ng;
    out->nslots = nslots;
    out->sig = tree_signature(out->groups.data(), ng, &out->sig2);
    out->topo = tree_topology(out->groups.data(), ng, nslots);
    return RTBHIP_OK;
}

Tree::~Tree() { for (auto &kv : dev_groups) (void)hipFree(kv.second); }

}  // namespace rtbhip
