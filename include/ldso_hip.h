/*
 * ldso_hip.h — C-ABI of libldso_hip.so: the MI355X (gfx950) implementation of LDSO's windowed
 * photometric bundle-adjustment hot path and CoarseTracker image alignment.
 *
 * The reference has no FFI layer; its boundary is two C++ member functions:
 *     float FullSystem::optimize(int mnumOptIts)                       src/frontend/FullSystem.cc:725
 *     bool  CoarseTracker::trackNewestCoarse(fh, SE3&, AffLight&, int, Vec5)   src/frontend/CoarseTracker.cc:61
 * Each entry point below names the reference function it replaces.  All functions are extern "C",
 * take plain pointers and sizes (layouts: ldso_window.h), return 0 on success or a negative
 * ldso_status code, never throw, and never free caller memory.  Handles are not thread-safe; distinct
 * handles are independent (one BA handle on the mapping thread, tracker handles on the tracking
 * thread, as FullSystem uses them).  Non-finite results are reported as LDSO_E_NONFINITE, the
 * analogue of the reference's isLost path (FullSystem.cc:845-849).
 */
#ifndef LDSO_HIP_H_
#define LDSO_HIP_H_

#include "ldso_window.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum ldso_status {
    LDSO_OK = 0,
    LDSO_E_INVALID = -1,      /* bad argument / window exceeds the handle's capacity            */
    LDSO_E_HIP = -2,          /* a HIP runtime call failed (see ldso_last_error)                */
    LDSO_E_NONFINITE = -3,    /* NaN/Inf in energies or the solve (reference: isLost = true)    */
    LDSO_E_UNSUPPORTED = -4,  /* a setting_* mode this implementation does not provide          */
    LDSO_E_NODEVICE = -5      /* no HIP device visible                                          */
};

int ldso_version(void);
const char *ldso_last_error(void);
int ldso_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Host-side helpers mirroring small reference functions that stay on the CPU.
 * ---------------------------------------------------------------------------------------------- */
/* FrameHessian::setEvalPT (FrameHessian.h:104-109) + setStateZero (FrameHessian.cc:12-42): fills
 * worldToCam_evalPT, state, state_zero and the numeric nullspaces of *f. */
int ldso_frame_set_evalPT(ldso_frame_t *f, const double worldToCam[12], const double state[10]);
/* FrameHessian::getPrior (FrameHessian.h:129-154): writes f->prior from f->frameID and the settings. */
int ldso_frame_set_prior(ldso_frame_t *f, const ldso_settings_t *s);
/* Defaults of src/Setting.cc for every field of ldso_settings_t. */
int ldso_settings_default(ldso_settings_t *s);
/* setGlobalCalib's pyramid rule (GlobalCalib.cc:20-29). */
int ldso_pyr_levels_used(int w, int h);

/* ------------------------------------------------------------------------------------------------
 * FrameHessian::dIp resident in HBM: one image pyramid per frame, built on the device from the raw irradiance
 * (FrameHessian::makeImages, FrameHessian.cc:44-113: level l >= 1 is the 2x2 mean of level l-1, pixels are (I, dx, dy) with
 * central-difference gradients) and shared BY POINTER between the coarse tracker (CoarseTracker.cc:248-256, :636-642: fh->dIp[lvl]),
 * the immature-point tracer (ImmaturePoint.cc:89: frame->dI) and the bundle adjustment (Residuals.cc:84: target->dI) - the
 * reference shares the same host arrays the same way.  4 bytes per pixel cross PCIe once per frame; every consumer below is
 * zero-copy and stream-ordered after the build (hipStreamWaitEvent on the pyramid's event, no host synchronisation).  The caller keeps
 * a pyramid alive while a consumer refers to it (until the consumer's next set_* call for that role / slot).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ldso_pyramid ldso_pyramid_t;
int ldso_pyr_create(int device, int w, int h, int levels, ldso_pyramid_t **out);
int ldso_pyr_destroy(ldso_pyramid_t *p);
/* irradiance: w*h floats on the host / on the device; enqueued on hip_stream (NULL: the default stream) */
int ldso_pyr_make_images(ldso_pyramid_t *p, const float *irradiance, void *hip_stream);
int ldso_pyr_make_images_device(ldso_pyramid_t *p, const void *irradiance_dev, void *hip_stream);
/* device pointer and size of a level ((w>>lvl)*(h>>lvl)*3 floats); ldso_pyr_get_level: test fetch to the host */
int ldso_pyr_level(ldso_pyramid_t *p, int lvl, const void **dev_ptr, int *wl, int *hl);
int ldso_pyr_get_level(ldso_pyramid_t *p, int lvl, float *out);

/* ------------------------------------------------------------------------------------------------
 * Windowed bundle adjustment (EnergyFunctional + the optimisation slice of FullSystem).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ldso_ba ldso_ba_t;

/* Allocate device state for windows up to max_frames x max_points on HIP device `device`. */
int ldso_ba_create(int device, int w, int h, int max_frames, int max_points, ldso_ba_t **out);
int ldso_ba_destroy(ldso_ba_t *h);
/* Run all launches on the caller's hipStream_t (e.g. torch's current stream); NULL = internal stream. */
int ldso_ba_set_stream(ldso_ba_t *h, void *hip_stream);
int ldso_ba_set_settings(ldso_ba_t *h, const ldso_settings_t *s);

/* Upload FrameHessian::dIp[0] (w*h Vec3f AoS: I,dx,dy; FrameHessian.h:169-175) into image slot
 * `slot` (0..max_frames-1).  Done once per new keyframe; slots are recycled on marginalisation. */
int ldso_ba_set_image(ldso_ba_t *h, int slot, const float *dI_level0_host);
/* Same, the pyramid level already being device-resident (zero-copy hand-over, not retained after the
 * next ldso_ba_set_image* on that slot).  A resident window that uses the slot follows the new buffer. */
int ldso_ba_set_image_device(ldso_ba_t *h, int slot, const void *dI_level0_dev);
/* The same slot from the RAW level-0 irradiance (w*h floats): FrameHessian::makeImages level 0 (FrameHessian.cc:44-113: copy +
 * central-difference gradients) runs on the device, 4 instead of 12 bytes per pixel cross PCIe.  ldso_ba_get_image: test fetch. */
int ldso_ba_set_image_raw(ldso_ba_t *h, int slot, const float *irradiance);
/* The slot from level 0 of a resident ldso_pyramid_t (zero-copy, as ldso_ba_set_image_device, ordered after the pyramid's build). */
int ldso_ba_set_image_pyramid(ldso_ba_t *h, int slot, ldso_pyramid_t *pyr);
int ldso_ba_get_image(ldso_ba_t *h, int slot, float *out_w_h_3);

/* K-splits (workgroups) per 16 x 16 tile of the Schur complement in the GN fast path (the device counterpart of AccumulatedSCHessianSSE's per-thread accumulators,
 * AccumulatedSCHessian.cc:53-119: each split owns a range of points and adds its fp32 partial tile into the fp64 system).  8 for a lone window (latency), 4 for the
 * windows of a batch of four or more (throughput; LDSO_BATCH_KS overrides).  Two runs agree bit for bit only under the same number. */
int ldso_ba_set_reduce_splits(ldso_ba_t *h, int splits);
/* Describe the window: EnergyFunctional::frames / allPoints / p->residuals after makeIDX
 * (EnergyFunctional.cc:380-401).  image_slot[f] = slot holding frame f's image.  linJ / lin_res_toZeroF
 * (R entries, may be NULL) are read where residuals[i].is_linearized != 0. */
int ldso_ba_set_window(ldso_ba_t *h, int F, const int32_t *image_slot, int P, const ldso_point_t *points,
                       int R, const ldso_residual_t *residuals, const ldso_rawjac_t *linJ, const float *lin_res_toZeroF);
/* The window of the next optimize() as a DELTA against the resident one: what the reference's own maintenance calls do to the window between two key
 * frames, in one call on the order EnergyFunctional::makeIDX produces (EnergyFunctional.cc: insertFrame :32, insertResidual :26, dropResidual :63,
 * removePoint :153, dropPointsF :224, marginalizeFrame :72, makeIDX :380).  Nothing of the surviving points crosses PCIe: their geometry, colours, inverse
 * depths, maxRelBaseline / numGoodResiduals and per-residual state / energy / activity are taken from the resident window on the device.
 *   frame_from[f]  old index of new frame f (old frames that appear nowhere were marginalised; survivors keep their order), -1 = insertFrame (last)
 *   point_from[i]  old row of new point i (old rows that appear nowhere were removed / dropped / marginalised; survivors keep their order),
 *                  -1 - k = the k-th fresh point (insertPoint), numbered in window order
 *   res_mask[i]    bit t: point i has a residual with target frame t (new numbering).  Set where the resident slot is empty = insertResidual (IN, energy 0,
 *                  isNew); clear where it is occupied = dropResidual
 *   fresh[n_fresh], fresh_res[n_fresh_res], fresh_mrb, fresh_ngr   the new points as in ldso_ba_set_window (host = new frame index), their residuals
 *                  point-major and target-ascending with .point = index into fresh, PointHessian::maxRelBaseline / numGoodResiduals
 * Flat residual order of the new window: point-major, target-ascending.  The result equals a fresh ldso_ba_set_window + ldso_ba_set_point_stats of the same
 * objects byte for byte (tests/test_resident_gpu.py); ldso_ba_set_frames / ldso_ba_set_prior follow as after ldso_ba_set_window.  Windows with linearised
 * residuals, shards and the first window of a handle go through ldso_ba_set_window. */
int ldso_ba_update_window(ldso_ba_t *h, int F, const int32_t *image_slot, const int32_t *frame_from, int P, const int32_t *point_from, const uint32_t *res_mask,
                          int n_fresh, const ldso_point_t *fresh, int n_fresh_res, const ldso_residual_t *fresh_res, const float *fresh_mrb,
                          const int32_t *fresh_ngr);
/* The same delta recorded call by call, as the reference edits its window (EnergyFunctional.cc; host-side bookkeeping until the commit, which applies the
 * edit as ONE ldso_ba_update_window and rejects an invalid one without touching the resident window).  Frames and residual targets are named by their index
 * in the RESIDENT window (frames inserted during the edit: the id ldso_ba_insert_frame returns), points by their resident row:
 *   ldso_ba_remove_frame    EnergyFunctional::marginalizeFrame :72 / FullSystem::marginalizeFrame FullSystem.cc:602-645 - the frame leaves, with the residuals
 *                           that target it and the points it hosts (the prior's Schur complement is ldso_ba_marginalize_frame)
 *   ldso_ba_insert_frame    insertFrame :32
 *   ldso_ba_remove_points   removePoint :153, dropPointsF :224, the points marginalizePointsF :165 absorbed
 *   ldso_ba_drop_residuals  dropResidual :63          ldso_ba_add_residuals   insertResidual :26 (IN, energy 0, isNew)
 *   ldso_ba_add_points      insertPoint + its residuals: n points in the layout of ldso_ba_set_window (host / target = frame ids of this edit, residuals[].point =
 *                           index into the call's points), each placed in front of resident row before_row[i] (number of rows = at the end): makeIDX :380 order
 *   ldso_ba_window_commit   surviving frames keep their order, inserted frames follow */
int ldso_ba_window_begin(ldso_ba_t *h);
int ldso_ba_remove_frame(ldso_ba_t *h, int frame_idx);
int ldso_ba_insert_frame(ldso_ba_t *h, int image_slot, int *frame_id_out);
int ldso_ba_remove_points(ldso_ba_t *h, int n, const int32_t *rows);
int ldso_ba_drop_residuals(ldso_ba_t *h, int n, const int32_t *rows, const int32_t *targets);
int ldso_ba_add_residuals(ldso_ba_t *h, int n, const int32_t *rows, const int32_t *targets);
int ldso_ba_add_points(ldso_ba_t *h, int n, const ldso_point_t *points, const int32_t *before_row, int n_res, const ldso_residual_t *residuals,
                       const float *max_rel_baseline, const int32_t *num_good_residuals);
int ldso_ba_window_commit(ldso_ba_t *h);
/* the dimensions of the window the handle holds (what the getters below write: R residual records, P point records, F frames); any pointer may be NULL */
int ldso_ba_get_dims(ldso_ba_t *h, int *F, int *P, int *R);
/* Frame / calibration state (FrameHessian::setState..., CalibHessian::setValue) and the
 * marginalisation prior HM,bM ((8F+4)^2 row-major, (8F+4); NULL = zero).  ldso_ba_set_frames also performs
 * EnergyFunctional::setAdjointsF (EnergyFunctional.cc:431-489) and FullSystem::setPrecalcValues
 * (FullSystem.cc:1423-1431) on the device.  Lifetime of the prior: ldso_ba_set_window resets it to zero (the
 * dimension 8F+4 changes with the window); ldso_ba_set_frames leaves it alone, so the two calls below may
 * come in either order after ldso_ba_set_window. */
int ldso_ba_set_frames(ldso_ba_t *h, const ldso_frame_t *frames, const ldso_calib_t *calib);
int ldso_ba_set_prior(ldso_ba_t *h, const double *HM, const double *bM);
/* Optional, after ldso_ba_set_window: PointHessian::maxRelBaseline / numGoodResiduals of the P points as the reference's objects hold them
 * (they persist across optimize() calls: FullSystem.cc:1521-1536, AccumulatedSCHessian.cc:14-21).  Without the call both start at zero. */
int ldso_ba_set_point_stats(ldso_ba_t *h, const float *maxRelBaseline, const int32_t *numGoodResiduals);

/* Multi-GPU: this rank owns points [begin,end) of the window (whole points only).
 * reduce_buf_dev is a caller-allocated device buffer of ldso_ba_reduce_doubles() doubles that the
 * caller all-reduces (sum) between ldso_ba_reduce_local() and ldso_ba_solve_reduced(). */
int ldso_ba_set_shard(ldso_ba_t *h, int point_begin, int point_end);
/* Points per workgroup of the fused linearisation kernel: 0 (default) = as few as keep one window's grid within one workgroup per CU (the
 * latency of a single window); n > 0 (multiple of 4) = fixed chunks of n points (ldso_ba_batch_create applies its own to the windows of a
 * batch: many windows fill the launch, a workgroup then takes several points per wavefront).  The fp32 partial sums of the top Hessian
 * are formed per chunk: two handles agree bit for bit only under the same chunking. */
int ldso_ba_set_chunk_points(ldso_ba_t *h, int points_per_workgroup);
int ldso_ba_get_chunk_points(ldso_ba_t *h, int *points_per_workgroup, int *workgroups);
/* The chunks themselves: ends[i] = one past the last point of chunk i (ascending, the last one = number of points; chunks never straddle a host frame -
 * a cut is added at every host boundary).  ldso_ba_batch_create cuts the windows of a batch UNEVENLY (every workgroup of the batched launch gets the same
 * amount of work, round 6); ldso_ba_get_chunk_cuts returns the cuts in force (n_out = their number; ends may be NULL to ask for the count),
 * ldso_ba_set_chunk_cuts applies them to another handle holding a window of the same shape (n = 0: back to ldso_ba_set_chunk_points' policy). */
int ldso_ba_get_chunk_cuts(ldso_ba_t *h, int32_t *ends, int cap, int *n_out);
int ldso_ba_set_chunk_cuts(ldso_ba_t *h, const int32_t *ends, int n);
size_t ldso_ba_reduce_doubles(ldso_ba_t *h);

/* --- the optimisation slice, one entry per reference function ------------------------------------ */
/* FullSystem::optimize preamble (FullSystem.cc:735-755): activeResiduals = !isLinearized, resetOOB. */
int ldso_ba_collect_active(ldso_ba_t *h);
/* FullSystem::linearizeAll(fix) (FullSystem.cc:1442-1492) = PointFrameResidual::linearize on every
 * active residual (Residuals.cc:13-214) + setNewFrameEnergyTH (:1762-1793); with fix != 0 also
 * applyRes, maxRelBaseline/numGoodResiduals and removal flags.  energy_out = Vec3[0]. */
int ldso_ba_linearize_all(ldso_ba_t *h, int fixLinearization, double *energy_out);
/* FullSystem::applyRes_Reductor(true) (FullSystem.cc:1706-1709) -> PointFrameResidual::applyRes. */
int ldso_ba_apply_res(ldso_ba_t *h);
/* FullSystem::backupState (FullSystem.cc:1625-1673, non-momentum branch). */
int ldso_ba_backup_state(ldso_ba_t *h);
/* FullSystem::solveSystem (FullSystem.cc:1433-1440) -> EnergyFunctional::solveSystemF
 * (EnergyFunctional.cc:240-351): accumulate A/L/SC, stitch, solve, orthogonalise, resubstitute. */
int ldso_ba_solve_system(ldso_ba_t *h, int iteration, double lambda);
/* FullSystem::doStepFromBackup(1,1,1,1,1) (FullSystem.cc:1546-1623) incl. setPrecalcValues. */
int ldso_ba_do_step(ldso_ba_t *h, int *canbreak_out);
/* FullSystem::loadSateBackup (FullSystem.cc:1675-1692). */
int ldso_ba_load_state_backup(ldso_ba_t *h);
/* EnergyFunctional::calcMEnergyF and calcLEnergyF_MT (EnergyFunctional.cc:353-378, 627-682) at the current state: the prior /
 * linearised-residual energies of the LM accept test (FullSystem.cc:805-826).  ldso_ba_optimize uses them when
 * settings.forceAcceptStep == 0 (accept / loadSateBackup + lambda *= 100, one host round trip per stage). */
int ldso_ba_calc_lm_energies(ldso_ba_t *h, double *energy_M, double *energy_L);
/* FullSystem::optimize(mnumOptIts) (FullSystem.cc:725-864) with everything on the device and no host
 * synchronisation inside the loop.  force_all_iterations != 0 ignores `canbreak` (BASELINE config C3).
 * rmse_out = the function's return value; iterations_out = GN iterations executed. */
int ldso_ba_optimize(ldso_ba_t *h, int mnumOptIts, int force_all_iterations, float *rmse_out, int *iterations_out);
/* EnergyFunctional::marginalizePointsF (EnergyFunctional.cc:165-222) for the points with flags[p] != 0 (the caller's
 * PS_MARGINALIZED decision, FullSystem.cc:1208-1270), fused with the reset + re-linearise + fixLinearizationF pass that
 * FullSystem::flagPointsForRemoval runs on exactly these points (FullSystem.cc:1241-1250, Residuals.cc:216-242):
 * HM += margWeightFac (M - Msc), bM += margWeightFac (Mb - Mbsc) on the device (and copied to HM_out / bM_out when
 * non-NULL, (8F+4)^2 row-major and 8F+4 doubles).  The window's applied state is unchanged; the caller removes the
 * points (next ldso_ba_set_window) and calls EnergyFunctional::marginalizeFrame on the returned prior as before. */
int ldso_ba_marginalize_points(ldso_ba_t *h, const int32_t *flags, double *HM_out, double *bM_out);
/* EnergyFunctional::marginalizeFrame (EnergyFunctional.cc:72-151) on the device prior H_M / b_M (as set by
 * ldso_ba_set_prior or left by ldso_ba_marginalize_points) with the frame's prior / delta_prior: out = prior of the window
 * without frame `frame_idx`, (8(F-1)+4)^2 row-major and 8(F-1)+4 doubles. */
int ldso_ba_marginalize_frame(ldso_ba_t *h, int frame_idx, double *HM_out, double *bM_out);
/* Point activation: FullSystem::optimizeImmaturePoint (FullSystem.cc:892-1010; ImmaturePoint::linearizeResidual,
 * ImmaturePoint.cc:312-381) for n immature points against the key frames of the resident window (its images, calibration and
 * CURRENT poses, i.e. after ldso_ba_set_frames): up to gn_iterations LM steps on the inverse depth from the middle of
 * [idepth_min, idepth_max]; min_obs = 1, min_idepth_hessian = setting_minIdepthH_act (100), gn_iterations =
 * setting_GNItsOnPointActivation (3) in the reference (FullSystem.cc:1204).  out[i].ok says whether the point is created;
 * out[i].res_state[t] == 0 lists the target frames that get a PointFrameResidual. */
int ldso_ba_activate_points(ldso_ba_t *h, int n, const ldso_immature_t *points, int min_obs, float min_idepth_hessian, int gn_iterations,
                            ldso_activation_t *out);
/* Asynchronous variant used by bench.py: enqueue `iters` Gauss-Newton iterations (solveSystem +
 * doStepFromBackup + linearizeAll + applyRes) on the handle's stream and return immediately.  The launch sequence of a call is captured into a HIP graph the
 * first time and replayed when the same call comes again (same window, settings, stream, first iteration and count: the key is a hash of every launch
 * argument); LDSO_GN_GRAPHS=0 in the environment at handle creation: launch by launch.  Same kernels, same arguments, same order either way. */
int ldso_ba_enqueue_gn(ldso_ba_t *h, int first_iteration, int iters);
int ldso_ba_sync(ldso_ba_t *h);

/* multi-GPU split of solveSystemF: local accumulate+stitch into the reduce buffer, then (after the
 * caller's all-reduce) the replicated solve + step + precalc. */
/* Multi-GPU fast path (what bench.py --gpus N runs): the all-reduce buffer IS the HFinal / bFinal accumulator of the
 * 3-launch iteration.  Layout, ldso_ba_gn_reduce_doubles() doubles: [HFinal lower triangle (8F+4)^2 | bFinal 8F+4 |
 * 8 scalar sums | P newest-frame energy candidates (value+1)].  Per iteration: ldso_ba_gn_reduce_local (rank-local
 * accumulate; the H_M / prior / lambda terms are added by the rank that owns point 0) -> all-reduce(sum) by the caller
 * -> ldso_ba_gn_solve_reduced (replicated solve + step + linearizeAll of the shard + applyRes).  lambda as solveSystemF. */
size_t ldso_ba_gn_reduce_doubles(ldso_ba_t *h);
int ldso_ba_gn_reduce_local(ldso_ba_t *h, void *reduce_buf_dev, double lambda);
int ldso_ba_gn_solve_reduced(ldso_ba_t *h, const void *reduce_buf_dev, int iteration, double lambda);
/* Batched windows (one 7-keyframe window is tiny for an MI355X): the forced Gauss-Newton iteration of n
 * INDEPENDENT windows (several agents, sequences or hypotheses) with three launches per iteration for the whole batch
 * (k_reduce_batch -> k_gn_solve_batch -> k_linearize_batch).  Per-window arithmetic is that of ldso_ba_enqueue_gn.  The handles
 * share one device and one stream, hold windows of the same slot-table width (all F <= 8 or all F in 9..16) without linearised
 * residuals, and have been brought to the same stage (window set, ldso_ba_linearize_all + ldso_ba_apply_res).  Results are
 * fetched per handle as usual. */
typedef struct ldso_ba_batch ldso_ba_batch_t;
int ldso_ba_batch_create(ldso_ba_t *const *handles, int n, ldso_ba_batch_t **out);
int ldso_ba_batch_reduce_splits(ldso_ba_batch_t *b, int *splits);          /* K-splits per Schur tile the batch reduces its windows with (ldso_ba_set_reduce_splits) */
/* The chunking ldso_ba_batch_create applies, as host logic without a device (tests/test_batch_balance_cpu.py): n_seg runs of points that may share a chunk (one window's
 * points of one host frame each, in launch order), n_wg workgroups, a chunk costing chunk_cost points on top of its own -> chunk_end[] (cumulative point counts, ascending,
 * a chunk never spans two segments) and wg_first_chunk[n_wg + 1] (workgroup w works through the chunks [wg_first_chunk[w], wg_first_chunk[w + 1])), chosen so that the
 * largest workgroup load (points + chunk_cost per chunk) is minimal for this greedy cut (bisection on the budget, *budget_out).  Returns the number of chunks. */
int ldso_ba_balance_chunks(int n_seg, const int32_t *seg_points, int n_wg, int chunk_cost, int32_t *chunk_end, int cap, int32_t *wg_first_chunk, int64_t *budget_out);
int ldso_ba_batch_enqueue_gn(ldso_ba_batch_t *b, int first_iteration, int iters);
int ldso_ba_batch_destroy(ldso_ba_batch_t *b);
/* points per workgroup ldso_ba_batch_create gave the windows of the batch - since round 6 the AVERAGE, rounded: the cuts are uneven, ldso_ba_get_chunk_cuts has
 * them - (0: their single-window chunking was kept; ldso_ba_batch_destroy restores it) */
int ldso_ba_batch_chunk_points(ldso_ba_batch_t *b, int *points_per_workgroup);
/* bench / profiling: average duration [us] of the batched k_linearize over `reps` back-to-back launches (HIP events on the batch's stream) */
int ldso_ba_batch_time_linearize(ldso_ba_batch_t *b, int reps, double *avg_us);
/* The same sharded iteration with the collective inside, for a C / C++ host (north_star: host code stays C++): `iters` forced
 * Gauss-Newton iterations of this rank's shard - k_reduce into the handle's all-reduce buffer, ncclAllReduce (RCCL, fp64 sum, in
 * place, ~29 KB + 8 P bytes at F = 7) on the handle's stream, replicated solve, linearize of the shard - enqueued without a host
 * synchronisation.  `nccl_comm` is an ncclComm_t created by the caller (one rank per GPU, ncclCommInitRank); ncclAllReduce is
 * resolved at run time from the RCCL already in the process, else from librccl.so.  Every rank calls it with the same arguments. */
int ldso_ba_enqueue_gn_rccl(ldso_ba_t *h, void *nccl_comm, int first_iteration, int iters);
/* The same sharded iteration with a ONE-SHOT PEER-WRITE all-reduce instead of RCCL's ring (the message is 29 KB + 8 P bytes,
 * latency-bound): every rank owns a receive window (2 parities x n_ranks slots) that its peers address over xGMI; a rank writes its partial
 * into its slot of EVERY window as self-validating 64-bit words (payload | exchange number: no fence across the fabric) and sums the slots
 * of its own window in rank order (deterministic).  ldso_ba_p2p_window_alloc: this rank's window (uncached device memory) and, optionally,
 * its hipIpcMemHandle_t (64 bytes) for a peer PROCESS, which maps it with ldso_ba_p2p_window_open; ranks inside one process (or with peer
 * access enabled) pass the pointers themselves.  windows[q] = rank q's window as this process addresses it, windows[rank] = the own one.
 * Every rank calls ldso_ba_enqueue_gn_p2p with the same (first_iteration, iters).  ldso_ba_p2p_check (after ldso_ba_sync): LDSO_E_HIP when
 * a peer's words did not arrive within the kernel's 2 s poll limit.
 * The slots of a window are laid out by the CAPACITY of the handle (ldso_ba_create's max_frames / max_points), not by the current window: all ranks
 * create their handles with the same capacities, and a window change (ldso_ba_set_window) between two exchanges moves nothing.  Where uncached
 * device memory cannot be had ldso_ba_p2p_window_alloc fails with LDSO_E_UNSUPPORTED (cached memory would never show a polling kernel its peers' stores).
 * The exchange number lives in the HANDLE (1, 2, 3 ... over all ldso_ba_enqueue_gn_p2p calls of its lifetime) and must advance in lock-step
 * on all ranks: the handles and windows of the n_ranks ranks form ONE generation.  When any rank re-creates its handle, every rank re-creates
 * its handle and re-allocates its window (alloc zeroes it) - words of an older generation carrying the same number would otherwise be taken
 * for the new partial, numbers that never meet end in the 2 s timeout. */
size_t ldso_ba_p2p_window_bytes(ldso_ba_t *h, int n_ranks);
int ldso_ba_p2p_window_alloc(ldso_ba_t *h, int n_ranks, void **window_out, void *ipc_handle_out_64_bytes);
int ldso_ba_p2p_window_open(ldso_ba_t *h, const void *ipc_handle_64_bytes, void **window_out);
int ldso_ba_p2p_window_close(ldso_ba_t *h, void *window, int opened_from_handle);
int ldso_ba_enqueue_gn_p2p(ldso_ba_t *h, int rank, int n_ranks, void *const *windows, int first_iteration, int iters);
int ldso_ba_p2p_check(ldso_ba_t *h);
int ldso_ba_reduce_local(ldso_ba_t *h, void *reduce_buf_dev);
int ldso_ba_solve_reduced(ldso_ba_t *h, const void *reduce_buf_dev, int iteration, double lambda, int do_step);

/* --- results --------------------------------------------------------------------------------------- */
/* Per-residual outputs in the caller's flat residual order (any pointer may be NULL). */
int ldso_ba_get_residuals(ldso_ba_t *h, ldso_res_out_t *out, int32_t *state_state, int32_t *is_active, int32_t *to_remove);
int ldso_ba_get_points(ldso_ba_t *h, ldso_point_out_t *out);
int ldso_ba_get_frames(ldso_ba_t *h, ldso_frame_t *frames, double *step /*F*10*/, double *calib_value, double *calib_step,
                       double *pre_worldToCam /*F*12*/);
/* ldso_ba_get_residuals + ldso_ba_get_points + ldso_ba_get_frames behind ONE stream synchronisation (what FullSystem::optimize's tail reads back in one go:
 * FullSystem.cc:815-864).  res / state_state / is_active / to_remove / frames / step / calib_* may be NULL, points must not; the outputs equal the three getters'. */
int ldso_ba_get_results(ldso_ba_t *h, ldso_res_out_t *res, int32_t *state_state, int32_t *is_active, int32_t *to_remove, ldso_point_out_t *points,
                        ldso_frame_t *frames, double *step /*F*10*/, double *calib_value, double *calib_step);
/* Stitched systems of the last solve, (8F+4)^2 / (8F+4) doubles, reference ordering [calib | frames]. */
int ldso_ba_get_system(ldso_ba_t *h, double *HA, double *bA, double *HL, double *bL, double *Hsc, double *bsc,
                       double *HFinal, double *bFinal, double *x);
/* RawResidualJacobian of selected residuals (for fixLinearizationF callers): recomputed on demand at the
 * current state, or - with ldso_ba_set_debug_dump(h,1) - the copy written by the last linearize pass. */
int ldso_ba_set_debug_dump(ldso_ba_t *h, int enable);
/* Debug / test: run the reduce and the control step of a fast-path iteration as two launches (k_reduce, k_gn_solve) even where
 * the fused k_reduce_solve launch applies - the two schedules must agree (tests/test_ba_gpu.py::test_fused_launch_stress). */
int ldso_ba_set_debug_split_launch(ldso_ba_t *h, int enable);
int ldso_ba_get_jacobians(ldso_ba_t *h, const int32_t *res_ids, int n, ldso_rawjac_t *out);
/* Pair precalc [h*F+t][27] as FrameFramePrecalc: KRKi 9, Kt 3, R0 9, t0 3, aff 2, b0 1. */
int ldso_ba_get_precalc(ldso_ba_t *h, float *out);
/* Pair transforms of the CURRENT state [h*F+t][14]: PRE_RTll 9, PRE_tTll 3, PRE_aff_mode 2 (what point activation reads). */
int ldso_ba_get_pair_rt(ldso_ba_t *h, float *out);
int ldso_ba_get_counts(ldso_ba_t *h, int *resInA, int *resInL);
/* energies of the linearizeAll calls inside the last ldso_ba_optimize (<= cap values), returns count */
int ldso_ba_get_energy_log(ldso_ba_t *h, double *out, int cap);
/* bench.py roofline: average duration (us) of `reps` back-to-back launches of the dominant kernel (k_linearize, applied state
 * -> scratch set, nothing applied) between one pair of HIP events on the handle's stream. */
int ldso_ba_time_linearize(ldso_ba_t *h, int reps, double *avg_us);
/* average device time (ms) per launch of kernel `which` since the last reset (HIP events on the
 * handle's stream); which: 0 = linearize, 1 = reduce+gather, 2 = solve, 3 = point step, 4 = an empty event pair
 * recorded right after every linearize launch (the event overhead to subtract). */
int ldso_ba_kernel_time_ms(ldso_ba_t *h, int which, double *avg_ms, int *launches);
int ldso_ba_profile(ldso_ba_t *h, int enable);

/* ------------------------------------------------------------------------------------------------
 * CoarseTracker (src/frontend/CoarseTracker.cc).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ldso_tracker ldso_tracker_t;

int ldso_tr_create(int device, int w, int h, int levels, ldso_tracker_t **out);
int ldso_tr_destroy(ldso_tracker_t *t);
int ldso_tr_set_stream(ldso_tracker_t *t, void *hip_stream);
int ldso_tr_set_settings(ldso_tracker_t *t, const ldso_settings_t *s);
/* CoarseTracker::makeK (CoarseTracker.cc:219-246). */
int ldso_tr_make_k(ldso_tracker_t *t, const ldso_calib_t *calib);
/* CoarseTracker::setCoarseTrackingRef (CoarseTracker.cc:248-256) + makeCoarseDepthL0 (:258-438).
 * ref_dIp: `levels` host pointers to the reference keyframe pyramid; pts = n x (Ku,Kv,new_idepth,HdiF)
 * of the active points whose lastResiduals[0] is IN, in the reference's iteration order. */
int ldso_tr_set_ref(ldso_tracker_t *t, const float *const *ref_dIp, float ref_aff_a, float ref_aff_b, float ref_exposure,
                    const float *pts, int n);
/* the frame to be tracked (FrameHessian::dIp[0..levels-1]) */
int ldso_tr_set_new_frame(ldso_tracker_t *t, const float *const *new_dIp, float exposure);
/* The new frame from its RAW level-0 irradiance (w*h floats): the whole FrameHessian::makeImages pyramid (2x2 mean pooling +
 * gradients per level, FrameHessian.cc:44-113) is built on the device.  ldso_tr_get_new_frame_level: test fetch of one level. */
int ldso_tr_set_new_frame_image(ldso_tracker_t *t, const float *irradiance, float exposure);
int ldso_tr_get_new_frame_level(ldso_tracker_t *t, int lvl, float *out);
/* Both frames as resident ldso_pyramid_t (zero-copy): the frame that was tracked becomes the next reference without any copy. */
int ldso_tr_set_new_frame_pyramid(ldso_tracker_t *t, ldso_pyramid_t *pyr, float exposure);
int ldso_tr_set_ref_pyramid(ldso_tracker_t *t, ldso_pyramid_t *pyr, float ref_aff_a, float ref_aff_b, float ref_exposure,
                            const float *pts, int n);
/* CoarseTracker::calcRes (CoarseTracker.cc:440-572). rs_out = Vec6; returns buf_warped_n via n_warped. */
int ldso_tr_calc_res(ldso_tracker_t *t, int lvl, const double T_ref2new[12], float aff_a, float aff_b, float cutoffTH,
                     double rs_out[6], int *n_warped);
/* CoarseTracker::calcGSSSE (CoarseTracker.cc:574-632) for the buffers of the last calcRes. */
int ldso_tr_calc_gs(ldso_tracker_t *t, int lvl, const double T_ref2new[12], float aff_a, float aff_b, double H_out[64], double b_out[8]);
/* CoarseTracker::trackNewestCoarse (CoarseTracker.cc:61-217): whole LM pyramid loop on the device. */
int ldso_tr_track(ldso_tracker_t *t, double T_ref2new_inout[12], float aff_inout[2], int coarsestLvl, const double minResForAbort[5],
                  double lastResiduals_out[5], double lastFlowIndicators_out[3], int *ok_out, int *iterations_out);
/* evaluate nhyp motion hypotheses of FullSystem::trackNewCoarse (FullSystem.cc:213-357)
 * concurrently, one workgroup each (nhyp <= 128).  Arrays are nhyp-major. */
int ldso_tr_track_batch(ldso_tracker_t *t, int nhyp, double *T_ref2new_inout, float *aff_inout, int coarsestLvl, const double minResForAbort[5],
                        double *lastResiduals_out, double *lastFlowIndicators_out, int *ok_out, int *iterations_out);
/* The hypothesis loop of FullSystem::trackNewCoarse (FullSystem.cc:319-356) replayed on the results of one
 * ldso_tr_track_batch call that ran every try with minRes = NaN: reproduces the sequential accept / abort / early-exit decisions
 * (a try counts as aborted at the level where its residual exceeds 1.5 x the `achievedRes` of the tries before it).  Host only.
 * best_out = index of the winning try or -1; tries_consumed_out = how many tries the sequential loop would have run. */
/* calcRes evaluations per pyramid level of the last track (hypothesis 0 of a batch) and the reference point counts pc_n[lvl] */
int ldso_tr_last_track_evals(ldso_tracker_t *t, int evals[5], int pc_n[5]);
/* LM solves of the last track call (all hypotheses) whose 8 x 8 system was rank-deficient to float precision (a pivot below 1e-6 of the diagonal entry it started from: fewer than
 * eight reference points on a level) and were solved by the reference's diagonally pivoted LDL^T (Eigen's `Hl.ldlt().solve(-b)`, CoarseTracker.cc:120-128) instead of the unpivoted register
 * factorisation; 0 on any normal track. */
int ldso_tr_last_track_pivoted_solves(ldso_tracker_t *t, int *n);
/* Debug / test entry: the 8 x 8 LM solve of the tracking kernel on its own - H with diag_scale (= 1 + lambda) on the diagonal, x = -b (`Hl.ldlt().solve(-b)`,
 * CoarseTracker.cc:120-128); *pivoted = 1 when the system went through the pivoted factorisation.  Runs on the current device, synchronous. */
int ldso_tr_debug_solve8(const double H[64], const double b[8], double diag_scale, double x[8], int *pivoted);
int ldso_tr_select_hypothesis(int nhyp, int coarsestLvl, const double *lastResiduals /*nhyp*5*/, const int *ok /*nhyp*/, double lastCoarseRMSE0,
                              double reTrackThreshold, int *best_out, int *tries_consumed_out, double achievedRes_out[5]);
/* The motion-hypothesis list of FullSystem::trackNewCoarse (FullSystem.cc:189-309): worldToCam poses [R|t] (Frame::getPose()) of
 * allFrameHistory[size-3] (sprelast), allFrameHistory[size-2] (slast) and coarseTracker->lastRef (lastF) -> up to 83 lastF_2_fh tries
 * (out: 83 x 12 doubles).  poses_valid = 0 (one of the three poses invalid, :306-309): the identity alone.  Pure host function. */
int ldso_tr_motion_hypotheses(const double sprelast_w2c[12], const double slast_w2c[12], const double lastF_w2c[12], int poses_valid,
                              double *lastF_2_fh_tries_out, int *n_out);
/* Vec4 FullSystem::trackNewCoarse(fh) (FullSystem.cc:179-386) on a tracker whose reference and new frame are set: hypothesis list, the try
 * loop with its achievedRes abort thresholds and reTrackThreshold early exit (try 0 alone; the remaining tries, if the loop goes on, as one
 * batched launch + ldso_tr_select_hypothesis), the pose / affine hand-over.  lastCoarseRMSE: FullSystem::lastCoarseRMSE in / out;
 * result4 = (achievedRes[0], flowVecs); new_frame_w2c / aff_out = what :376-378 write into fh->frame; *good = 0: "tracking failed entirely". */
int ldso_tr_track_new_coarse(ldso_tracker_t *t, const double sprelast_w2c[12], const double slast_w2c[12], const double lastF_w2c[12], int poses_valid,
                             const float aff_last[2], double lastCoarseRMSE_inout[5], double reTrackThreshold, double result4[4],
                             double new_frame_w2c[12], float aff_out[2], int *tries_consumed, int *good);
int ldso_tr_get_pc(ldso_tracker_t *t, int lvl, float *u, float *v, float *idepth, float *color, int *n);

/* ------------------------------------------------------------------------------------------------------------
 * Immature-point tracing: FullSystem::traceNewCoarse (FullSystem.cc:1012-1050) = ImmaturePoint::traceOn
 * (src/internal/ImmaturePoint.cc:47-310) for every immature point of the window against a new frame, one launch.
 * The points stay resident on the device between frames (set once per key frame, traced on every frame).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct ldso_tracer ldso_tracer_t;
int ldso_trace_settings_default(ldso_trace_settings_t *s);
int ldso_trace_create(int device, int w, int h, int max_points, ldso_tracer_t **out);
int ldso_trace_destroy(ldso_tracer_t *t);
int ldso_trace_set_settings(ldso_tracer_t *t, const ldso_trace_settings_t *s);
int ldso_trace_set_points(ldso_tracer_t *t, int n, const ldso_immature_t *points);
int ldso_trace_get_points(ldso_tracer_t *t, ldso_immature_t *points_out);
/* the new frame: level-0 image as FrameHessian::dIp[0] (w*h*3 floats), or the raw irradiance (makeImages on the device) */
int ldso_trace_set_frame(ldso_tracer_t *t, const float *dI_level0);
int ldso_trace_set_frame_raw(ldso_tracer_t *t, const float *irradiance);
int ldso_trace_set_frame_pyramid(ldso_tracer_t *t, ldso_pyramid_t *pyr);
/* per host key frame h < n_hosts (FullSystem.cc:1025-1032): KRKi[h] = K R K^-1 (row-major 3x3), Kt[h] = K t,
 * aff[h] = AffLight::fromToVecExposure(host, new).  counts_out[6] (optional): points per resulting LDSO_IPS_* status. */
int ldso_trace_on(ldso_tracer_t *t, int n_hosts, const float *KRKi, const float *Kt, const float *aff, int *counts_out);

/* ------------------------------------------------------------------------------------------------------------
 * Monocular initialiser: CoarseInitializer (src/frontend/CoarseInitializer.cc, include/frontend/CoarseInitializer.h).
 * Replaces setFirst (:547-619, minus the pixel selection and the kd-tree of makeNN, whose results arrive as the point records),
 * trackFrame (:40-178) and what it calls: calcResAndGS (:181-405), calcEC (:412-428), optReg (:430-459), propagateUp/Down
 * (:462-522), resetPoints (:621-643), doStep (:645-671), applyStep (:673-687), makeK (:689-715).
 * The whole Levenberg-Marquardt loop of one trackFrame runs on the device without a host round trip.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct ldso_initializer ldso_initializer_t;
int ldso_init_create(int device, int w, int h, int pyr_levels, ldso_initializer_t **out);
int ldso_init_destroy(ldso_initializer_t *t);
int ldso_init_set_stream(ldso_initializer_t *t, void *hip_stream);
/* makeK + setFirst: calib = CalibHessian fxl,fyl,cxl,cyl of level 0; the first frame as raw irradiance (w*h floats, makeImages on
 * the device) with its exposure; points[lvl] / n_points[lvl]: the selected pixels of every level with neighbours and parents. */
int ldso_init_set_first(ldso_initializer_t *t, const float calib[4], const float *irradiance, float ab_exposure,
                        const ldso_init_point_t *const *points, const int *n_points, float huberTH, int fixAffine);
/* trackFrame(newFrame): returns the state after the call in state_out (optional) */
int ldso_init_track_frame(ldso_initializer_t *t, const float *irradiance, float ab_exposure, ldso_init_state_t *state_out);
int ldso_init_get_state(ldso_initializer_t *t, ldso_init_state_t *state_out);
int ldso_init_set_state(ldso_initializer_t *t, const ldso_init_state_t *state);
int ldso_init_get_points(ldso_initializer_t *t, int lvl, ldso_init_point_t *points_out);
int ldso_init_set_points(ldso_initializer_t *t, int lvl, const ldso_init_point_t *points);
/* stage entry for parity tests: one calcResAndGS(lvl, refToNew, aff) on the current new frame; idepth_new of the points is used
 * as it stands.  H,Hsc 8x8 row-major, b,bsc 8, res = (E.A, alphaEnergy, E.num), ec = calcEC(lvl). */
int ldso_init_calc_res_and_gs(ldso_initializer_t *t, int lvl, const double refToNew[12], double aff_a, double aff_b,
                              float *H, float *b, float *Hsc, float *bsc, float *res, float *ec);
int ldso_init_set_new_frame(ldso_initializer_t *t, const float *irradiance, float ab_exposure);
/* Launch schedule of ldso_init_track_frame (tuning / debugging; the results do not depend on it, bit for bit: tests/test_init_gpu.py).
 * first_steps: control steps (evaluation + control launch) enqueued before the state is read back for the first time; the rest of the
 *   frame's maximum is enqueued only if the frame has not finished by then.  0 (default) = what the previous frame took + 25 % + 4 (first frame: half of the maximum).
 * prepare_on_grid: 1 (default) = the inputs of the optReg sweeps are prepared by a grid kernel between evaluation and control step
 *   once the initialiser has snapped; 0 = by the control block itself (one compute unit), as in the frame that snaps. */
int ldso_init_set_schedule(ldso_initializer_t *t, int first_steps, int prepare_on_grid);
/* Host logic, no device: the schedule of the in-place optReg sweep (CoarseInitializer.cc:430-459: points are updated in index order, reading neighbours that
 * may already have been updated) as ldso_init_set_first builds it for one level.  neighbours [n][10] (-1 = none), width = points per pass (32 on the device).
 * pass_out[i] = pass of point i: every neighbour j < i in an earlier pass, every neighbour j > i in the same or a later one, <= width points per pass.
 * Returns the number of passes, or a negative LDSO_E_* code. */
int ldso_init_sweep_schedule(int n, const int *neighbours, int width, int *pass_out);
/* debug builds (LDSO_STAMPS=1): accumulated device-side counters, zeros otherwise */
int ldso_init_debug_counters(ldso_initializer_t *t, long long out[8]);

#ifdef __cplusplus
}
#endif
#endif /* LDSO_HIP_H_ */
