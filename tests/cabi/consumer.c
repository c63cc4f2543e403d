/* tests/cabi/consumer.c -- TEST INFRASTRUCTURE: a plain C99 program on include/rtbhip.h and librtbhip.so, what a maintainer's cgo / JNI / ctypes
 * binding sits on.  `consumer symbols` takes the address of every entry point the header declares (link-time proof that the library exports them)
 * and calls the two that need no device; `consumer fkine N` builds the Panda of models/ETS/Panda.py through rtbhip_chain_create, evaluates
 * rtbhip_fkine_jacob on N host rows (q_k[j] = 0.1 (j + 1) + 1e-3 k) and prints a checksum the Python test recomputes through the oracle;
 * `consumer shard N` is the multi-GPU split (see shard_mode). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rtbhip.h"

/* symbols.inc is written by tests/test_c_consumer.py from the declarations of include/rtbhip.h: one X(name) per entry point */
typedef void (*anyfn)(void);
static const anyfn k_entry_points[] = {
#define X(f) (anyfn)f,
#include "symbols.inc"
#undef X
};

static void set_const(rtbhip_et *e, double tx, double ty, double tz, int rot_axis, double angle)
{
    double c = cos(angle), s = sin(angle);
    int k;
    memset(e, 0, sizeof *e);
    e->kind = RTBHIP_ET_CONST;
    for (k = 0; k < 16; k++) e->T[k] = (k % 5 == 0) ? 1.0 : 0.0;
    e->T[3] = tx; e->T[7] = ty; e->T[11] = tz;
    if (rot_axis == 0) { e->T[5] = c; e->T[6] = -s; e->T[9] = s; e->T[10] = c; }         /* Rx */
    if (rot_axis == 2) { e->T[0] = c; e->T[1] = -s; e->T[4] = s; e->T[5] = c; }          /* Rz */
}

/* models/ETS/Panda.py:32-54 through rtbhip_chain_create */
static int make_panda(rtbhip_chain_t *chain, int *n_ets, int *n_joints)
{
    const double h = 1.5707963267948966;
    int m = 0, joint = 0;
    rtbhip_et ets[22];
    #define CONST(tx, ty, tz, ax, ang) set_const(&ets[m++], tx, ty, tz, ax, ang)
    #define JOINT() do { memset(&ets[m], 0, sizeof ets[m]); ets[m].kind = RTBHIP_ET_RZ; ets[m].jindex = joint++; m++; } while (0)
    CONST(0, 0, 0.333, -1, 0); JOINT();
    CONST(0, 0, 0, 0, -h); JOINT();
    CONST(0, 0, 0, 0, h); CONST(0, 0, 0.316, -1, 0); JOINT();
    CONST(0.0825, 0, 0, -1, 0); CONST(0, 0, 0, 0, h); JOINT();
    CONST(-0.0825, 0, 0, -1, 0); CONST(0, 0, 0, 0, -h); CONST(0, 0, 0.384, -1, 0); JOINT();
    CONST(0, 0, 0, 0, h); JOINT();
    CONST(0.088, 0, 0, -1, 0); CONST(0, 0, 0, 0, h); CONST(0, 0, 0.107, -1, 0); JOINT();
    CONST(0, 0, 0.103, -1, 0); CONST(0, 0, 0, 2, -h / 2);
    #undef CONST
    #undef JOINT
    *n_ets = m; *n_joints = joint;
    return rtbhip_chain_create(ets, m, NULL, chain);
}

#define CHECK(call) do { if ((call) != RTBHIP_OK) { fprintf(stderr, "%s: %s\n", #call, rtbhip_last_error()); return 4; } } while (0)

/* `consumer shard N [p2p]`: the multi-GPU split of SURVEY 8e from plain C, ONE process driving every visible GPU -- world = the device count
 * (1 on a single-GPU box: a real world-size-1 RCCL communicator), one stream per device.  Rank r evaluates its rtbhip_shard_range rows with
 * rtbhip_fkine_jacob_packed into device memory; ONE rtbhip_shard_gather per rank (between rtbhip_shard_group(1) / (0)) brings the 464-byte
 * T||J rows to rank 0, a second one (root = -1) to every rank; the checksum of the gathered rows is printed for the Python test to recompute
 * through the oracle.  `p2p` forces the grouped send / receive form (rtbhip_tune "shard_p2p"). */
static int shard_mode(long N, int p2p)
{
    enum { MAXW = 16 };
    int world = 0, r, j, m, nj;
    int32_t cw = -1, cr = -1, ver = 0;
    long k;
    rtbhip_chain_t chain = 0;
    rtbhip_comm_t comms[MAXW];
    void *streams[MAXW], *dq[MAXW], *dtj[MAXW], *dall[MAXW], *droot = NULL;
    int64_t begin[MAXW], count[MAXW];
    double *q, *rows, *rows2, sum = 0.0;
    int same = 1;
    CHECK(rtbhip_device_count(&world));
    if (world < 1) { fprintf(stderr, "no device\n"); return 5; }
    if (world > MAXW) world = MAXW;
    CHECK(make_panda(&chain, &m, &nj));
    CHECK(rtbhip_tune("shard_p2p", p2p));
    CHECK(rtbhip_shard_comm_create_all(world, NULL, comms));
    CHECK(rtbhip_shard_comm_info(comms[world - 1], &cw, &cr, &ver));
    q = malloc(sizeof(double) * 7 * (size_t)N);
    for (k = 0; k < N; k++) for (j = 0; j < 7; j++) q[7 * k + j] = 0.1 * (j + 1) + 1e-3 * k;
    for (r = 0; r < world; r++) {
        CHECK(rtbhip_shard_range(N, r, world, &begin[r], &count[r]));
        CHECK(rtbhip_stream_create(r, &streams[r]));
        CHECK(rtbhip_device_alloc(r, (uint64_t)count[r] * 56, &dq[r]));
        CHECK(rtbhip_device_alloc(r, (uint64_t)count[r] * 464, &dtj[r]));
        CHECK(rtbhip_device_alloc(r, (uint64_t)N * 464, &dall[r]));
        CHECK(rtbhip_device_copy(dq[r], q + 7 * begin[r], (uint64_t)count[r] * 56, 1, streams[r]));
        CHECK(rtbhip_fkine_jacob_packed(chain, dq[r], count[r], NULL, NULL, 0, dtj[r], RTBHIP_MEM_DEVICE, streams[r]));
    }
    CHECK(rtbhip_device_alloc(0, (uint64_t)N * 464, &droot));
    CHECK(rtbhip_shard_group(1));                                            /* gather to rank 0 */
    for (r = 0; r < world; r++)
        CHECK(rtbhip_shard_gather(comms[r], dtj[r], count[r], 464, N, world, r, 0, r == 0 ? droot : NULL, streams[r]));
    CHECK(rtbhip_shard_group(0));
    CHECK(rtbhip_shard_group(1));                                            /* ... and to every rank */
    for (r = 0; r < world; r++)
        CHECK(rtbhip_shard_gather(comms[r], dtj[r], count[r], 464, N, world, r, -1, dall[r], streams[r]));
    CHECK(rtbhip_shard_group(0));
    for (r = 0; r < world; r++) CHECK(rtbhip_stream_sync(streams[r]));
    rows = malloc((size_t)N * 464); rows2 = malloc((size_t)N * 464);
    CHECK(rtbhip_device_copy(rows, droot, (uint64_t)N * 464, 2, NULL));
    for (r = 0; r < world; r++) {
        CHECK(rtbhip_device_copy(rows2, dall[r], (uint64_t)N * 464, 2, NULL));
        same = same && memcmp(rows, rows2, (size_t)N * 464) == 0;
    }
    for (k = 0; k < N; k++) {
        for (j = 0; j < 16; j++) sum += rows[58 * k + j] * (1 + j);
        for (j = 0; j < 42; j++) sum += rows[58 * k + 16 + j] * (1 + j);
    }
    printf("world %d comm_world %d comm_rank %d rccl %d rows %ld checksum %.12f allgather_equal %d p2p %d\n", world, (int)cw, (int)cr, (int)ver, N, sum, same, p2p);
    for (r = 0; r < world; r++) {
        CHECK(rtbhip_shard_comm_destroy(comms[r]));
        CHECK(rtbhip_device_free(dq[r])); CHECK(rtbhip_device_free(dtj[r])); CHECK(rtbhip_device_free(dall[r]));
        CHECK(rtbhip_stream_destroy(streams[r]));
    }
    CHECK(rtbhip_device_free(droot));
    rtbhip_chain_destroy(chain);
    free(q); free(rows); free(rows2);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "symbols") == 0) {
        int n = 0;
        size_t i;
        for (i = 0; i < sizeof k_entry_points / sizeof k_entry_points[0]; i++) n += k_entry_points[i] != (anyfn)0;
        printf("symbols %d version %d\n", n, rtbhip_version());
        return 0;
    }
    if (argc >= 3 && strcmp(argv[1], "fkine") == 0) {
        long N = atol(argv[2]), k;
        int j, m = 0, joint = 0;
        rtbhip_chain_t chain = 0;
        double *q, *T, *J, sum = 0.0;
        if (make_panda(&chain, &m, &joint) != RTBHIP_OK) { fprintf(stderr, "chain_create: %s\n", rtbhip_last_error()); return 2; }
        q = malloc(sizeof(double) * 7 * N); T = malloc(sizeof(double) * 16 * N); J = malloc(sizeof(double) * 42 * N);
        for (k = 0; k < N; k++) for (j = 0; j < 7; j++) q[7 * k + j] = 0.1 * (j + 1) + 1e-3 * k;
        if (rtbhip_fkine_jacob(chain, q, N, NULL, NULL, 0, T, J, RTBHIP_MEM_HOST, NULL) != RTBHIP_OK) { fprintf(stderr, "fkine_jacob: %s\n", rtbhip_last_error()); return 3; }
        for (k = 0; k < 16 * N; k++) sum += T[k] * (1 + (k % 16));
        for (k = 0; k < 42 * N; k++) sum += J[k] * (1 + (k % 42));
        printf("joints %d ets %d rows %ld checksum %.12f\n", joint, m, N, sum);
        rtbhip_chain_destroy(chain);
        free(q); free(T); free(J);
        return 0;
    }
    if (argc >= 3 && strcmp(argv[1], "shard") == 0) return shard_mode(atol(argv[2]), argc >= 4 && strcmp(argv[3], "p2p") == 0);
    fprintf(stderr, "usage: consumer symbols | consumer fkine N | consumer shard N [p2p]\n");
    return 1;
}
