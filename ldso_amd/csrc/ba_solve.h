// ba_solve.h — launch arguments of the control kernel (see ba_solve.hip)
#pragma once
#include <stdint.h>

enum {
    SK_POST = 1u << 0, SK_ADJ = 1u << 1, SK_GATHER = 1u << 2, SK_SOLVE = 1u << 3, SK_BACKUP = 1u << 4, SK_STEP = 1u << 5,
    SK_LOADBK = 1u << 6, SK_PRECALC = 1u << 7, SK_REANCHOR = 1u << 8, SK_COLLECT = 1u << 9, SK_LOG = 1u << 10,
    SK_FROMREDUCED = 1u << 11, SK_THRESH = 1u << 12, SK_EXPORT = 1u << 13,
    SK_NONULLSPACE = 1u << 14      // with SK_ADJ: adjoints only, keep the gauge nullspace basis (no solve follows)
};
enum { PS_RESUB = 1, PS_BACKUP = 2, PS_STEP = 4, PS_LOAD = 8 };

struct SolveArgs {
    unsigned flags;
    int iteration;
    double lambda;
    int hasL;
    int hasPrior;               // HM/bM non-zero
    int GSP;
    int logIdx;
    double *reduceOut;          // multi-GPU: rank-local sums are exported here (SK_GATHER)
    const double *reduceIn;     // multi-GPU: all-reduced sums are read from here (SK_FROMREDUCED)
    int itCheck;                // k_gn_solve: >= 0 = un-forced optimize(): iteration index for the device-side `canbreak` early exit
    int *waitCtr;               // k_reduce_solve: counter the reduce workgroups of the same launch increment when their sums are in B.acc
    int waitTarget;             //                 ... and its value when all of them are done (0 / nullptr: no wait)
    int *hostStop;              // un-forced optimize(): host-mapped word that receives the index of the iteration that ended the loop (canbreak, or the
    int lastIt;                 //                       last enqueued iteration lastIt) as soon as the device knows it - no stream synchronisation
};
