/* tests/cabi/consumer.c -- TEST INFRASTRUCTURE: a plain C99 program on include/rtbhip.h and librtbhip.so, what a maintainer's cgo / JNI / ctypes
 * binding sits on.  `consumer symbols` takes the address of every entry point the header declares (link-time proof that the library exports them)
 * and calls the two that need no device; `consumer fkine N` builds the Panda of models/ETS/Panda.py through rtbhip_chain_create, evaluates
 * rtbhip_fkine_jacob on N host rows (q_k[j] = 0.1 (j + 1) + 1e-3 k) and prints a checksum the Python test recomputes through the oracle. */
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
        const double h = 1.5707963267948966;
        long N = atol(argv[2]), k;
        int j, m = 0, joint = 0;
        rtbhip_et ets[22];
        rtbhip_chain_t chain = 0;
        double *q, *T, *J, sum = 0.0;
        /* models/ETS/Panda.py:32-54 */
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
        if (rtbhip_chain_create(ets, m, NULL, &chain) != RTBHIP_OK) { fprintf(stderr, "chain_create: %s\n", rtbhip_last_error()); return 2; }
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
    fprintf(stderr, "usage: consumer symbols | consumer fkine N\n");
    return 1;
}
