/*
 * ldso_window.h — plain-C data layouts that cross the drop-in boundary.
 *
 * These are the flattened (SoA-friendly) images of the reference's object graph for ONE sliding
 * window.  They carry no behaviour.  Every field cites the reference member it mirrors
 * (paths relative to the LDSO source tree).
 *
 * Index invariants (bit-exact with the reference):
 *   - frames[]  : window order, index == FrameHessian::idx (EnergyFunctional.cc:382-383).
 *   - points[]  : EnergyFunctional::allPoints order = frames x frame->features filtered by
 *                 VALID && ACTIVE (EnergyFunctional.cc:387-399).
 *   - residuals of point k : res[points[k].res_begin .. +res_count), in the order of
 *                 PointHessian::residuals (PointHessian.h:100).  FullSystem::activeResiduals
 *                 (FullSystem.cc:735-755) is this array filtered by !is_linearized.
 *   - accumulator slots: host + target*F; accD slot host + t1*F + t2*F*F; xAd slot host*F+target.
 */
#ifndef LDSO_WINDOW_H_
#define LDSO_WINDOW_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDSO_PATTERN_NUM 8   /* Settings.h:163  patternNum                         */
#define LDSO_CPARS 4         /* NumTypes.h      CPARS                              */
#define LDSO_MAX_FRAMES 16   /* capacity of this implementation (reference default window: 7) */
#define LDSO_PYR_LEVELS 6    /* Settings.h:8    PYR_LEVELS                         */

/* ResState, Residuals.h:32-34 */
enum { LDSO_RES_IN = 0, LDSO_RES_OOB = 1, LDSO_RES_OUTLIER = 2 };

/* solver mode bits, Settings.h:12-23 */
enum {
    LDSO_SOLVER_SVD = 1, LDSO_SOLVER_ORTHOGONALIZE_SYSTEM = 2, LDSO_SOLVER_ORTHOGONALIZE_POINTMARG = 4,
    LDSO_SOLVER_ORTHOGONALIZE_FULL = 8, LDSO_SOLVER_SVD_CUT7 = 16, LDSO_SOLVER_REMOVE_POSEPRIOR = 32,
    LDSO_SOLVER_USE_GN = 64, LDSO_SOLVER_FIX_LAMBDA = 128, LDSO_SOLVER_ORTHOGONALIZE_X = 256,
    LDSO_SOLVER_MOMENTUM = 512, LDSO_SOLVER_STEPMOMENTUM = 1024, LDSO_SOLVER_ORTHOGONALIZE_X_LATER = 2048
};

/* The subset of the global setting_* knobs the hot path reads (Settings.h / Setting.cc:8-130). */
typedef struct ldso_settings {
    float huberTH;                    /* setting_huberTH = 9                          Setting.cc:75  */
    float outlierTHSumComponent;      /* setting_outlierTHSumComponent = 50*50        Setting.cc:42  */
    float affineOptModeA;             /* setting_affineOptModeA = 1e12                Setting.cc:65  */
    float affineOptModeB;             /* setting_affineOptModeB = 1e8                 Setting.cc:66  */
    float frameEnergyTHN;             /* 0.7                                          Setting.cc:77  */
    float frameEnergyTHFacMedian;     /* 1.5                                          Setting.cc:79  */
    float frameEnergyTHConstWeight;   /* 0.5                                          Setting.cc:76  */
    float overallEnergyTHWeight;      /* 1                                            Setting.cc:80  */
    float initialCalibHessian;        /* 5e9                                          Setting.cc:22  */
    float margWeightFac;              /* 0.5*0.5                                      Setting.cc:45  */
    float idepthFixPriorMargFac;      /* 600*600                                      Setting.cc:17  */
    float thOptIterations;            /* 1.2                                          Setting.cc:38  */
    float coarseCutoffTH;             /* 20                                           Setting.cc:81  */
    int32_t minOptIterations;         /* 1                                            Setting.cc:37  */
    int32_t solverMode;               /* SOLVER_FIX_LAMBDA|SOLVER_ORTHOGONALIZE_X_LATER Setting.cc:23 */
    int32_t forceAcceptStep;          /* setting_forceAceptStep = true                Setting.cc:73  */
    double solverModeDelta;           /* 1e-5                                         Setting.cc:24  */
} ldso_settings_t;

/* One keyframe of the window: FrameHessian (FrameHessian.h:27-214). */
typedef struct ldso_frame {
    double worldToCam_evalPT[12];     /* SE3 as row-major [R|t] 3x4        FrameHessian.h:182 */
    double state[10];                 /* FrameHessian::state               FrameHessian.h:185 */
    double state_zero[10];            /* FrameHessian::state_zero          FrameHessian.h:191 */
    double prior[8];                  /* getPrior().head<8>()              FrameHessian.h:129-154, FrameHessian.cc:115 */
    double nullspaces_pose[36];       /* Mat66, row-major; column i = i-th pose nullspace  FrameHessian.cc:18-27 */
    double nullspaces_scale[6];       /* Vec6                              FrameHessian.cc:29-37 */
    double nullspaces_affine[8];      /* Mat42 row-major                   FrameHessian.cc:39-42 */
    float ab_exposure;                /* exposure time                     FrameHessian.h:176 */
    float frameEnergyTH;              /* dynamic outlier threshold         FrameHessian.h:175 */
    int32_t frameID;                  /* keyframe id (0 == first frame)    FrameHessian.h:159 */
    int32_t pad_;
} ldso_frame_t;

/* Camera intrinsics state: CalibHessian (CalibHessian.h:16-140). value = unscaled [fx,fy,cx,cy]/SCALE. */
typedef struct ldso_calib {
    double value[4];                  /* CalibHessian::value       */
    double value_zero[4];             /* CalibHessian::value_zero  */
} ldso_calib_t;

/* One active point: PointHessian (PointHessian.h:83-131). */
typedef struct ldso_point {
    float u, v;                       /* pixel position in host                      */
    float idepth;                     /* == idepth_scaled (SCALE_IDEPTH = 1)         */
    float idepth_zero;                /* == idepth_zero_scaled                       */
    float color[LDSO_PATTERN_NUM];    /* host colours of the 8 pattern pixels        */
    float weights[LDSO_PATTERN_NUM];  /* host gradient weights                       */
    float priorF;                     /* PointHessian::takeData(): hasDepthPrior ? setting_idepthFixPrior : 0 */
    int32_t host;                     /* host frame idx in window                    */
    int32_t res_begin;                /* CSR into the residual array                 */
    int32_t res_count;
} ldso_point_t;

/* One photometric residual: PointFrameResidual (Residuals.h:40-130). */
typedef struct ldso_residual {
    int32_t point;                    /* index into points[]                         */
    int32_t host;                     /* hostIDX                                     */
    int32_t target;                   /* targetIDX                                   */
    int32_t state_state;              /* ResState                                    */
    int32_t is_linearized;            /* isLinearized                                */
    int32_t is_active;                /* isActiveAndIsGoodNEW                        */
    int32_t is_new;                   /* isNew                                       */
    float state_energy;               /* state_energy                                */
} ldso_residual_t;

/* RawResidualJacobian (RawResidualJacobian.h:13-39), 74 floats, field order as declared there. */
typedef struct ldso_rawjac {
    float resF[8];
    float Jpdxi[2][6];
    float Jpdc[2][4];
    float Jpdd[2];
    float JIdx[2][8];
    float JabF[2][8];
    float JIdx2[4];                   /* Mat22f row-major */
    float JabJIdx[4];
    float Jab2[4];
} ldso_rawjac_t;

/* Per-residual result of linearize() (Residuals.cc:13-214 outputs kept on the object). */
typedef struct ldso_res_out {
    float state_NewEnergy;
    float state_NewEnergyWithOutlier;
    int32_t state_NewState;
    float centerProjectedTo[3];
    float JpJdF[8];                   /* after applyRes/takeData (Residuals.h:123-128) */
} ldso_res_out_t;

/* Per-point result of the Schur step (AccumulatedSCHessian.cc:9-51, EnergyFunctional.cc:518-547). */
typedef struct ldso_point_out {
    float step;
    float HdiF;
    float bdSumF;
    float idepth_hessian;
    float Hdd_accAF, bd_accAF, Hcd_accAF[4];
    float Hdd_accLF, bd_accLF, Hcd_accLF[4];
    float idepth;                     /* current idepth after the last doStepFromBackup */
    float maxRelBaseline;
    int32_t numGoodResiduals;
} ldso_point_out_t;

/* ------------------------------------------------------------------------------------------------------------
 * Immature points (first half): ImmaturePoint::traceOn / FullSystem::traceNewCoarse.
 * One record = the members of ldso::internal::ImmaturePoint that traceOn reads and writes
 * (include/internal/ImmaturePoint.h:60-125); `host` = index of the host key frame in the per-host pose arrays.
 * ------------------------------------------------------------------------------------------------------------ */
enum { LDSO_IPS_GOOD = 0, LDSO_IPS_OOB = 1, LDSO_IPS_OUTLIER = 2, LDSO_IPS_SKIPPED = 3, LDSO_IPS_BADCONDITION = 4, LDSO_IPS_UNINITIALIZED = 5 };

typedef struct ldso_immature {
    float u, v;                       /* feature->uv */
    float color[8], weights[8];       /* ImmaturePoint ctor (ImmaturePoint.cc:21-36) */
    float gradH[4];                   /* sum over the pattern of grad grad^T, row-major 2x2 */
    float energyTH;                   /* patternNum * setting_outlierTH * overallEnergyTHWeight^2 */
    float idepth_min, idepth_max;     /* in / out; idepth_max = NaN: unbounded */
    float quality;                    /* in / out (second-best / best energy of the discrete search) */
    int32_t lastTraceStatus;          /* in / out, LDSO_IPS_* */
    float lastTraceUV[2];             /* out */
    float lastTracePixelInterval;     /* out */
    int32_t host;
    int32_t pad_;
} ldso_immature_t;                    /* 128 bytes */

typedef struct ldso_trace_settings {
    float maxPixSearch;               /* 0.027   Setting.cc:28 */
    float trace_stepsize;             /* 1.0     Setting.cc:89 */
    float trace_GNThreshold;          /* 0.1     Setting.cc:91 */
    float trace_extraSlackOnTH;       /* 1.2     Setting.cc:92 */
    float trace_slackInterval;        /* 1.5     Setting.cc:93 */
    float trace_minImprovementFactor; /* 2       Setting.cc:94 */
    float huberTH;                    /* 9       Setting.cc:76 */
    int32_t trace_GNIterations;       /* 3       Setting.cc:90 */
    int32_t minTraceTestRadius;       /* 2       Setting.cc:52 */
    int32_t pad_;
} ldso_trace_settings_t;              /* 40 bytes */

/* Result of FullSystem::optimizeImmaturePoint for one immature point (FullSystem.cc:892-1010). */
typedef struct ldso_activation {
    float idepth;                     /* currentIdepth after the LM iterations (setIdepth / setIdepthZero of the new point) */
    int32_t ok;                       /* 1: a PointHessian is created; 0: rejected (returns nullptr / 0 in the reference) */
    int32_t numGoodRes;               /* residuals with state IN */
    int32_t iterations;               /* LM iterations executed */
    float energy;                     /* lastEnergy */
    float Hdd, bd;                    /* lastHdd, lastbd */
    int32_t pad_;
    int32_t res_state[LDSO_MAX_FRAMES];   /* per target frame idx: final state_state (0 IN, 1 OOB, 2 OUTLIER), -1 for the host */
} ldso_activation_t;                  /* 96 bytes */

/* One candidate of the monocular initialiser: struct Pnt (include/frontend/CoarseInitializer.h:19-57), bools widened to int32. */
typedef struct ldso_init_point {
    float u, v;                       /* pixel position on its pyramid level (x + 0.1, y + 0.1; CoarseInitializer.cc:578-579) */
    float idepth;
    int32_t isGood;
    float energy[2];                  /* (photometric, regulariser) */
    int32_t isGood_new;
    float idepth_new;
    float energy_new[2];
    float iR;                         /* regularised inverse depth */
    float iRSumNum;
    float lastHessian, lastHessian_new;
    float maxstep;
    int32_t parent;                   /* index on level + 1, -1 on the coarsest level */
    float parentDist;
    int32_t neighbours[10];           /* 10 nearest points of the same level (makeNN, CoarseInitializer.cc:717-783), -1: none */
    float neighboursDist[10];
    float my_type;
    float outlierTH;                  /* patternNum * setting_outlierTH (CoarseInitializer.cc:597) */
    float pad_;
} ldso_init_point_t;                  /* 160 bytes */

/* State of a CoarseInitializer between two trackFrame calls (CoarseInitializer.h:66-74,91-92). */
typedef struct ldso_init_state {
    double thisToNext[12];            /* row-major [R|t] */
    double aff_a, aff_b;              /* thisToNext_aff */
    int32_t snapped, snappedAt, frameID;
    int32_t ready;                    /* return value of the last trackFrame: snapped && frameID > snappedAt + 5 */
    int32_t evals;                    /* calcResAndGS evaluations of the last trackFrame */
    int32_t pad_;
} ldso_init_state_t;                  /* 136 bytes */

#ifdef __cplusplus
}
#endif
#endif /* LDSO_WINDOW_H_ */
