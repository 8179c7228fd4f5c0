// bh_kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels for the BEnlsip hot path.
//
// Data layout in HBM: the Jacobian block of this rank is stored ROW-MAJOR,
// rows padded to a multiple of 16 doubles (128 B): Jd[i*ld + j].  The host hands J over
// column-major (Julia); bh_hess_create transposes it once on the device.  Rows [d, d+q) of
// the same image hold C, so H*p = J'(Jp) + C'(mu C p) is ONE sweep with a per-row weight.
// Why row-major: n (<= 16384) is small enough that a whole row of J lives in the registers
// of one workgroup (n/T doubles per lane), so J'(Jp) needs ONE read of J: the row is dotted
// with p (wave64 DPP + permlane reduction, cross-wave through LDS), then scaled by that dot
// and accumulated into a register-resident slice of z — 8*d*n bytes instead of 16*d*n.
//
// Reference call sites: src/basic_tralcnlss.jl:92-106 (vthv, *), :690-764 (projected_cg),
// :793-809 (factor_to_boundary); src/polyhedral_constraints.jl:72-136 (left_mul, left_mul_tr,
// projection_nullspace!, projection_subspace!).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bh_reduce.hip.h"
#include "bh_matvec.hip.h"
#include "bh_cg.hip.h"
#include "bh_comm.hip.h"
#include "bh_cgfuse.hip.h"
#include "bh_proj.hip.h"
#include "bh_cauchy.hip.h"
#include "bh_minor.hip.h"
