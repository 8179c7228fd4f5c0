# BEnlsipHIP.jl — glue that routes BEnlsip.jl's trust-region subproblem hot path to the MI355X library
# (include/benlsip_hip.h) WITHOUT editing the package: it only adds more specific Float64 methods next to the
# parametric ones of src/basic_tralcnlss.jl and src/polyhedral_constraints.jl, so dispatch prefers them.
#
#   using BEnlsip; include("julia/BEnlsipHIP.jl"); using .BEnlsipHIP
#   BEnlsipHIP.init()                       # once per process (one process per GPU)
#   x, y = tralcnllss(x0, r, jac_r, c, jac_c, A, b, x_l, x_u)   # unchanged outer iteration
#
# What it shadows (all Float64 methods next to the package's parametric ones):
#   hot path          Base.:*(H, v), vthv, projection!, projected_cg                         (src/basic_tralcnlss.jl:92-106, :690-764; poly:158-170)
#   callers           minor_iterate (:649-675); opt-in: cauchy_step (device_cauchy_step!), inner_step with device-resident
#                     vectors (resident_inner_step!) (:394-460, :574-639)
#   row-shard seams   new_point, evaluate_al, first_derivatives (+ second_derivatives), least_squares_multipliers
#                     (:32-85, :887-903): mx and g all-reduced so that a multi-rank run stays in lock-step
#   opt-in            update_chol! (skip the host factor), reference_projection_form!
#
# STATUS: written against the reference at /root/reference and syntax-reviewed only.  No Julia toolchain exists in
# the build image or on the GPU box, so this file has never been executed there; the executable stand-in is the
# Python mirror benlsip.jl_amd/operators.py driving the same C ABI (tests/test_parity_gpu.py).
module BEnlsipHIP

using BEnlsip, LinearAlgebra

const libbh = get(ENV, "BENLSIP_HIP_LIB", "libbenlsip_hip.so")

function check(rc::Int32, what::AbstractString)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:bh_strerror, libbh), Cstring, (Int32,), rc))
    det = unsafe_string(ccall((:bh_last_error_detail, libbh), Cstring, ()))
    error("$what: $msg ($rc): $det")      # reference conventions replaced: AssertionError / PosDefException
end

"""Select the GPU (default: LOCAL_RANK or 0).  `flags = 1` records hipEvents around H*p launches (bh_stats)."""
init(device::Integer = parse(Int, get(ENV, "LOCAL_RANK", "0")); flags::Integer = 0) =
    check(ccall((:bh_init, libbh), Int32, (Int32, Int32), device, flags), "bh_init")

# ---- AlHessian (src/basic_tralcnlss.jl:6-10): device image cached per OBJECT -----------------------------------------
# A new AlHessian is constructed at every J change (:46, :84); keying on the object (not on the J array, which a user
# jac_res may reuse) makes that the upload point.
mutable struct HessHandle
    ptr::Ptr{Cvoid}
end
const HESS = WeakKeyDict{BEnlsip.AlHessian{Float64},HessHandle}()

function handle(H::BEnlsip.AlHessian{Float64})
    h = get!(HESS, H) do
        d, n = size(H.J)
        q = size(H.C, 1)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bh_hess_create, libbh), Int32,
                    (Ref{Ptr{Cvoid}}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Float64),
                    ref, H.J, d, n, max(stride(H.J, 2), 1), H.C, q, max(stride(H.C, 2), 1), H.mu), "bh_hess_create")
        hh = HessHandle(ref[])
        finalizer(hh) do x                     # idempotent: least_squares_multipliers finalizes its throw-away image early
            x.ptr == C_NULL || ccall((:bh_hess_destroy, libbh), Int32, (Ptr{Cvoid},), x.ptr)
            x.ptr = C_NULL
        end
        hh
    end
    check(ccall((:bh_hess_set_mu, libbh), Int32, (Ptr{Cvoid}, Float64), h.ptr, H.mu), "bh_hess_set_mu")
    return h.ptr
end

# Base.:*(H, v) — src/basic_tralcnlss.jl:102-106
function Base.:*(H::BEnlsip.AlHessian{Float64}, v::Vector{Float64})
    out = Vector{Float64}(undef, size(H.J, 2))
    check(ccall((:bh_hmul, libbh), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), handle(H), v, out), "bh_hmul")
    return out
end

# vthv(H, v) — src/basic_tralcnlss.jl:92-96
function BEnlsip.vthv(H::BEnlsip.AlHessian{Float64}, v::Vector{Float64})
    out = Ref{Float64}(0.0)
    check(ccall((:bh_vthv, libbh), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), handle(H), v, out), "bh_vthv")
    return out[]
end

# ---- MixedConstraints (src/polyhedral_constraints.jl:1-7) -----------------------------------------------------------
mutable struct ProjHandle
    ptr::Ptr{Cvoid}
end
const PROJ = WeakKeyDict{BEnlsip.MixedConstraints{Float64},ProjHandle}()

function handle(lincons::BEnlsip.MixedConstraints{Float64})
    h = get!(PROJ, lincons) do
        mA, n = size(lincons.lineq)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:bh_proj_create, libbh), Int32, (Ref{Ptr{Cvoid}}, Ptr{Float64}, Int64, Int64, Int64),
                    ref, lincons.lineq, mA, n, max(stride(lincons.lineq, 2), 1)), "bh_proj_create")
        hh = ProjHandle(ref[])
        finalizer(x -> ccall((:bh_proj_destroy, libbh), Int32, (Ptr{Cvoid},), x.ptr), hh)
        hh
    end
    # fixvars / chol are mutated between CG calls (active_bounds!, add_active!, update_chol! — :62-68, :203-261):
    # re-push on every use (n/8 + 8*mpp^2 bytes).  Only the lower triangle is read by the library; the factor built by
    # cholesky_aug_aat has uplo == 'L' (factors = L), the initial cholesky(A*A') has uplo == 'U' (SURVEY.md §0.3-15).
    mA, n = size(lincons.lineq)
    mpp = mA + count(lincons.fixvars)
    if !REFERENCE_PROJECTION_FORM[]
        # library default (reduced form): the device factors A_free*A_free' itself, so only the BitVector image travels
        # (n/8 bytes; an identical push is recognised by the library and costs nothing) — lincons.chol is never read
        check(ccall((:bh_proj_set_active, libbh), Int32,
                    (Ptr{Cvoid}, Ptr{UInt64}, Int64, Ptr{Float64}, Int64, Int64),
                    h.ptr, lincons.fixvars.chunks, n, C_NULL, mpp, max(mpp, 1)), "bh_proj_set_active")
    else
        L = lincons.chol.uplo == 'L' ? lincons.chol.factors : Matrix(lincons.chol.L)
        check(ccall((:bh_proj_set_active, libbh), Int32,
                    (Ptr{Cvoid}, Ptr{UInt64}, Int64, Ptr{Float64}, Int64, Int64),
                    h.ptr, lincons.fixvars.chunks, n, L, mpp, max(stride(L, 2), 1)), "bh_proj_set_active")
    end
    return h.ptr
end

# projection!(lincons, r, v) — src/polyhedral_constraints.jl:158-170  (projection(lincons, r) :150-155 calls it)
function BEnlsip.projection!(lincons::BEnlsip.MixedConstraints{Float64}, r::Vector{Float64}, v::Vector{Float64})
    check(ccall((:bh_project, libbh), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), handle(lincons), r, v), "bh_project")
    return
end

# ---- projected_cg (src/basic_tralcnlss.jl:690-764) ------------------------------------------------------------------
function BEnlsip.projected_cg(g_minor::Vector{Float64}, H::BEnlsip.AlHessian{Float64}, w_l::Vector{Float64},
                              w_u::Vector{Float64}, lincons::BEnlsip.MixedConstraints{Float64}, kappa2::Float64;
                              atol::Float64 = sqrt(eps(Float64)))
    n = length(g_minor)
    w = Vector{Float64}(undef, n)
    status = Ref{Int32}(-1); iters = Ref{Int32}(0); nh = Ref{Int32}(0)
    check(ccall((:bh_pcg, libbh), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64,
                 Ptr{Float64}, Ref{Int32}, Ref{Int32}, Ptr{Float64}, Int64, Ref{Int32}),
                handle(H), handle(lincons), g_minor, w_l, w_u, kappa2, atol, 1e-10,
                w, status, iters, C_NULL, 0, nh), "bh_pcg")
    # 0..3 = CG_status (:12); 4 = the reference's `nothing` (iterations exhausted or max_iter == 0, :753-761)
    return w, (status[] == 4 ? nothing : BEnlsip.CG_status(status[]))
end

# minor_iterate(x, s, g_model, H, lincons, delta, kappa2) — src/basic_tralcnlss.jl:649-675 — one device-resident call:
# step bounds (:660-665), projected_cg (:667), linesearch and scaling (:669-672); saves four PCIe round trips per minor iterate.
function BEnlsip.minor_iterate(x::Vector{Float64}, s::Vector{Float64}, g_model::Vector{Float64}, H::BEnlsip.AlHessian{Float64},
                               lincons::BEnlsip.MixedConstraints{Float64}, delta::Float64, kappa2::Float64)
    n = length(x)
    w = Vector{Float64}(undef, n)
    status = Ref{Int32}(-1); iters = Ref{Int32}(0); nh = Ref{Int32}(0); alpha = Ref{Float64}(0.0)
    check(ccall((:bh_minor_iterate, libbh), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64,
                 Float64, Float64, Ptr{Float64}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Float64}),
                handle(H), handle(lincons), x, s, g_model, lincons.xlow, lincons.xupp, delta, kappa2,
                sqrt(eps(Float64)), 1e-10, w, status, iters, nh, alpha), "bh_minor_iterate")
    return w, (status[] == 4 ? nothing : BEnlsip.CG_status(status[]))
end

# cauchy_step(x, g, H, chol_aat, lincons, delta) — src/basic_tralcnlss.jl:574-639 — device-resident breakpoint search.  The
# library leaves its own active set at the final one; lincons is brought to the state the reference's method leaves
# behind (fixvars + one refreshed factor instead of one O(p^3) rebuild per breakpoint).
# OPT-IN since round 3 (BEnlsipHIP.device_cauchy_step!(true)): the default configuration shadows exactly the rows of SURVEY.md
# §8(a) — H*v, vthv, projection!, projected_cg, minor_iterate — and leaves the reference's own cauchy_step in charge, whose H*d
# and projection! calls still reach the device one by one.  Reason: near a critical point the search direction P(-g) cancels
# 1e7 .. 1e9 of its digits and the breakpoint sequence is decided by rounding — in the reference's own arithmetic too
# (tests/test_oracle_cpu.py::test_cauchy_search_is_multimodal_on_the_pinned_operands) — so a second implementation of the
# search legitimately lands on another active set there; a drop-in should not add that source of divergence unasked.
const DEVICE_CAUCHY_STEP = Ref(false)
device_cauchy_step!(flag::Bool = true) = (DEVICE_CAUCHY_STEP[] = flag)

function BEnlsip.cauchy_step(x::Vector{Float64}, g::Vector{Float64}, H::BEnlsip.AlHessian{Float64},
                             chol_aat::Cholesky{Float64,Matrix{Float64}}, lincons::BEnlsip.MixedConstraints{Float64},
                             delta::Float64)
    if !DEVICE_CAUCHY_STEP[]
        # the reference's own method (more general signature); its H*d / projection! calls are routed to the device above
        return invoke(BEnlsip.cauchy_step, Tuple{Vector{T},Vector{T},BEnlsip.AlHessian{T},Cholesky{T,Matrix{T}},
                                                 BEnlsip.MixedConstraints{T},T} where T,
                      x, g, H, chol_aat, lincons, delta)
    end
    n = length(x)
    s_c = Vector{Float64}(undef, n)
    chunks = zeros(UInt64, length(lincons.fixvars.chunks))
    nbp = Ref{Int32}(0); nh = Ref{Int32}(0)
    check(ccall((:bh_cauchy_step, libbh), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64},
                 Ptr{UInt64}, Ref{Int32}, Ref{Int32}),
                handle(H), handle(lincons), x, g, lincons.xlow, lincons.xupp, delta, s_c, chunks, nbp, nh), "bh_cauchy_step")
    lincons.fixvars.chunks .= chunks
    BEnlsip.update_chol!(lincons, chol_aat)
    return s_c
end

# ---- optional: the whole inner step with its vectors resident in HBM (SURVEY.md §8 f-1 / f-2) -------------------------------
# inner_step(x, g, H, chol_aat, lincons, delta, nb_minor_step, kappa2, kappa3) — src/basic_tralcnlss.jl:394-460.  The reference's
# method moves s, w and g_minor through Julia arrays between every two calls; this method owns the loop and keeps them in HBM:
# x, g go up once, s comes down once, and between them only scalars and the n/8-byte BitVector image of the active set cross
# PCIe.  The active set grows on the device (bh_proj_update_active_dev: device-side active_bounds + a Gram downdate over the
# newly fixed columns instead of the O(p^3) cholesky_aug_aat rebuild).  Opt-in, because it replaces a driver-level method:
#     BEnlsipHIP.resident_inner_step!(true)
# Executable mirror: benlsip.jl_amd/operators.py::inner_step (tests/test_parity_gpu.py::test_inner_step_device_chain_against_oracle).
const RESIDENT_INNER_STEP = Ref(false)
resident_inner_step!(flag::Bool = true) = (RESIDENT_INNER_STEP[] = flag)

mutable struct DeviceVec
    ptr::Ptr{Cvoid}
end
function DeviceVec(n::Integer)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:bh_dev_alloc, libbh), Int32, (Ref{Ptr{Cvoid}}, Int64), ref, 8 * max(n, 1)), "bh_dev_alloc")
    dv = DeviceVec(ref[])
    finalizer(x -> ccall((:bh_dev_free, libbh), Int32, (Ptr{Cvoid},), x.ptr), dv)
    return dv
end
upload!(dv::DeviceVec, v::Vector{Float64}) = check(ccall((:bh_dev_upload, libbh), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64), dv.ptr, v, 8 * length(v)), "bh_dev_upload")
download!(v::Vector{Float64}, dv::DeviceVec) = check(ccall((:bh_dev_download, libbh), Int32, (Ptr{Float64}, Ptr{Cvoid}, Int64), v, dv.ptr, 8 * length(v)), "bh_dev_download")

const RESIDENT = WeakKeyDict{BEnlsip.MixedConstraints{Float64},Dict{Symbol,DeviceVec}}()

function BEnlsip.inner_step(x::Vector{Float64}, g::Vector{Float64}, H::BEnlsip.AlHessian{Float64},
                            chol_aat::Cholesky{Float64,Matrix{Float64}}, lincons::BEnlsip.MixedConstraints{Float64},
                            delta::Float64, nb_minor_step::Int, kappa2::Float64, kappa3::Float64)
    if !RESIDENT_INNER_STEP[]
        # the reference's own method (more general signature), whose calls still reach the device one by one
        return invoke(BEnlsip.inner_step, Tuple{Vector{T},Vector{T},BEnlsip.AlHessian{T},Cholesky{T,Matrix{T}},
                                                BEnlsip.MixedConstraints{T},T,Int,T,T} where T,
                      x, g, H, chol_aat, lincons, delta, nb_minor_step, kappa2, kappa3)
    end
    (m, n) = size(lincons.lineq)
    dv = get!(RESIDENT, lincons) do
        d = Dict{Symbol,DeviceVec}(k => DeviceVec(n) for k in (:x, :g, :s, :w, :gm, :xlow, :xupp))
        upload!(d[:xlow], lincons.xlow); upload!(d[:xupp], lincons.xupp)
        d
    end
    upload!(dv[:x], x); upload!(dv[:g], g)
    hH, hP = handle(H), handle(lincons)
    chunks = lincons.fixvars.chunks                                   # written in place: lincons.fixvars follows the device
    nbp = Ref{Int32}(0); nh = Ref{Int32}(0)
    check(ccall((:bh_cauchy_step_dev, libbh), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{UInt64}, Ref{Int32}, Ref{Int32}),
                hH, hP, dv[:x].ptr, dv[:g].ptr, dv[:xlow].ptr, dv[:xupp].ptr, delta, dv[:s].ptr, chunks, nbp, nh), "bh_cauchy_step_dev")   # :410
    check(ccall((:bh_hmul_add_dev, libbh), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), hH, dv[:s].ptr, dv[:g].ptr, dv[:gm].ptr),
          "bh_hmul_add_dev")                                                                                                          # :412
    red_norm(v::DeviceVec) = (o = Ref{Float64}(0.0);
                              check(ccall((:bh_reduced_gradient_norm_dev, libbh), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), hP, v.ptr, o),
                                    "bh_reduced_gradient_norm_dev"); o[])
    norm_reduced_g, norm_reduced_g_minor = red_norm(dv[:g]), red_norm(dv[:gm])                                                       # :420-421
    approx_solved = norm_reduced_g_minor <= kappa3 * norm_reduced_g
    max_minor_step = min(nb_minor_step, n - m - count(lincons.fixvars))                                                                # :425-426
    j = 1; cg_stop = false
    # inside this loop dv[:gm] holds H*s + g for the current s by construction: bh_step_accumulate_dev may add the H*w that the CG
    # loop of the preceding bh_minor_iterate_dev accumulated instead of sweeping J again (310 -> 7 us at config 3)
    check(ccall((:bh_set_option, libbh), Int32, (Cstring, Int64), "step_from_cg", 1), "bh_set_option(step_from_cg)")
    while j <= max_minor_step && !approx_solved && !cg_stop                                                                            # :430
        status = Ref{Int32}(-1); iters = Ref{Int32}(0); nhm = Ref{Int32}(0); alpha = Ref{Float64}(0.0)
        check(ccall((:bh_minor_iterate_dev, libbh), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Float64,
                     Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Float64}),
                    hH, hP, dv[:x].ptr, dv[:s].ptr, dv[:gm].ptr, dv[:xlow].ptr, dv[:xupp].ptr, delta, kappa2, sqrt(eps(Float64)), 1e-10,
                    dv[:w].ptr, status, iters, nhm, alpha), "bh_minor_iterate_dev")                                                    # :434
        cg_stop = status[] == 2                                       # negative_curvature
        check(ccall((:bh_step_accumulate_dev, libbh), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    hH, dv[:s].ptr, dv[:w].ptr, dv[:g].ptr, dv[:gm].ptr), "bh_step_accumulate_dev")                                  # :436-437
        n_at = Ref{Int32}(0); n_fix = Ref{Int32}(0); branch = Ref{Int32}(0)
        check(ccall((:bh_proj_update_active_dev, libbh), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ptr{UInt64}),
                    hP, dv[:x].ptr, dv[:s].ptr, dv[:xlow].ptr, dv[:xupp].ptr, delta, sqrt(eps(Float64)), n_at, n_fix, branch, chunks),
              "bh_proj_update_active_dev")                                                                                            # :439-453
        if branch[] == 0
            norm_reduced_g, norm_reduced_g_minor = red_norm(dv[:g]), red_norm(dv[:gm])                                               # :446-447
            approx_solved = norm_reduced_g_minor <= kappa3 * norm_reduced_g
        else
            approx_solved = true
        end
        j += 1
    end
    mr = Ref{Float64}(0.0)                    # still under the invariant: s'Hs = s.(g_minor - g) from the resident g_minor, no J*v sweep
    check(ccall((:bh_model_reduction_dev, libbh), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), hH, dv[:g].ptr, dv[:s].ptr, mr),
          "bh_model_reduction_dev")                                                                                                   # :458
    check(ccall((:bh_set_option, libbh), Int32, (Cstring, Int64), "step_from_cg", 0), "bh_set_option(step_from_cg)")
    s = Vector{Float64}(undef, n)
    download!(s, dv[:s])
    BEnlsip.update_chol!(lincons, chol_aat)       # one refresh of the host factor (a no-op under skip_host_factor!)
    return s, mr[]
end

# ---- optional: drop the host-side factor maintenance (SURVEY.md §8 f-1) ---------------------------------------------------
# With the library's default reduced projection form the device factors A_free*A_free' (mA x mA) itself and never reads
# lincons.chol.  The reference still rebuilds the augmented (mA+p) x (mA+p) factor from scratch in update_chol! after every
# active-set change (active_bounds!, add_active! — src/polyhedral_constraints.jl:62-68,203-261): O(p^3) on the host, the
# wall-clock bottleneck once H*p is fast.  Since every consumer of lincons.chol (projection!) is routed to the device, that
# rebuild can be skipped.  Opt-in, because it changes what lincons.chol holds for any other user code:
#     BEnlsipHIP.skip_host_factor!(true)
const SKIP_HOST_FACTOR = Ref(false)
skip_host_factor!(flag::Bool = true) = (SKIP_HOST_FACTOR[] = flag)

# The reference's augmented (mA+p) x (mA+p) projection form on the device (bh_set_option("proj_form", 0)): the caller's factor
# lincons.chol is then pushed with every active-set change (8*mpp^2 bytes over PCIe) and must be kept up to date on the host
# (incompatible with skip_host_factor!).  Off by default: the reduced form computes the same projector (DESIGN.md §4).
const REFERENCE_PROJECTION_FORM = Ref(false)
function reference_projection_form!(flag::Bool = true)
    flag && SKIP_HOST_FACTOR[] && error("reference_projection_form! needs the host factor: call skip_host_factor!(false) first")
    check(ccall((:bh_set_option, libbh), Int32, (Cstring, Int64), "proj_form", flag ? 0 : 1), "bh_set_option(proj_form)")
    REFERENCE_PROJECTION_FORM[] = flag
end

function BEnlsip.update_chol!(lincons::BEnlsip.MixedConstraints{Float64}, chol_aat::Cholesky{Float64,Matrix{Float64}})
    if SKIP_HOST_FACTOR[]
        return                                        # the device rebuilds its own reduced factor in bh_proj_set_active
    end
    lincons.chol = BEnlsip.cholesky_aug_aat(lincons.lineq, lincons.fixvars, chol_aat)    # the reference's body (:66)
    return
end

# ---- multi-GPU: one Julia process per GPU (e.g. MPI.jl / Distributed); rows of J and of r are sharded by the caller ----
# ENV["BH_COMM"] = "rccl" (default) | "ipc" (one-shot peer-buffer exchange, ranks of one node) | "both" before comm_init.
unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:bh_comm_unique_id, libbh), Int32, (Ptr{UInt8},), id), "bh_comm_unique_id"); id)
comm_init(rank::Integer, nranks::Integer, id::Vector{UInt8}) =
    check(ccall((:bh_comm_init, libbh), Int32, (Int32, Int32, Ptr{UInt8}), rank, nranks, id), "bh_comm_init")
comm_destroy() = check(ccall((:bh_comm_destroy, libbh), Int32, ()), "bh_comm_destroy")

# Under row sharding `residuals(x)` / `jac_res(x)` return THIS rank's rows.  Everything the driver computes from them with
# plain Matrix / Vector operations would then be a per-rank partial and the replicated control flow (rho = ared/pred,
# initial_tr(g), the multipliers) would diverge between ranks.  The reference touches residual rows directly in exactly
# four places; each gets a Float64 method that routes the row-dependent part through an all-reduced entry point
# (with one rank they compute the same values, on the device):
#     new_point                  src/basic_tralcnlss.jl:32-49   mx (:44) and g (:45)
#     evaluate_al                :51-61                         mx (:58)
#     first_derivatives          :63-77                         g (:74)
#     least_squares_multipliers  :887-903                       g = jac_res(x)' * residuals(x) (:893)
# (tralcnllss's own uses of rx — size(rx,1) and dot(rx,rx) at :214,:239,:291 — only feed the log.)
# These four are exercised, through the same C entry points, by tests/test_multirank_gpu.py::
# test_whole_row_sharded_solve_matches_unsharded_oracle (tests/hip_ops.py::ShardedHipOps is their executable mirror).

"dot(rx,rx) over ALL ranks' rows — bh_resid_sqnorm."
function resid_sqnorm(rx::Vector{Float64})
    out = Ref{Float64}(0.0)
    check(ccall((:bh_resid_sqnorm, libbh), Int32, (Ptr{Float64}, Int64, Ref{Float64}), rx, length(rx), out), "bh_resid_sqnorm")
    return out[]
end

"g = Jx'*rx + Cx'*y_bar with the J' product summed over all ranks (C'y_bar is added by rank 0) — bh_grad."
function gradient(H::BEnlsip.AlHessian{Float64}, rx::Vector{Float64}, y_bar::Vector{Float64})
    g = Vector{Float64}(undef, size(H.J, 2))
    check(ccall((:bh_grad, libbh), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), handle(H), rx, y_bar, g), "bh_grad")
    return g
end

function BEnlsip.new_point(x::Vector{Float64}, y::Vector{Float64}, mu::Float64, residuals::F1, nlconstraints::F2,
                           jac_res::F3, jac_nlcons::F4) where {F1<:Function, F2<:Function, F3<:Function, F4<:Function}
    rx, cx = residuals(x), nlconstraints(x)
    Jx, Cx = jac_res(x), jac_nlcons(x)
    y_bar = y + mu*cx
    mx = 0.5*resid_sqnorm(rx) + dot(y,cx) + 0.5*mu*dot(cx,cx)     # :44
    H = BEnlsip.AlHessian(Jx,Cx,mu)                                # :46 (first, so that g can use its device image)
    g = gradient(H, rx, y_bar)                                     # :45
    return rx, cx, y_bar, mx, g, H
end

function BEnlsip.evaluate_al(x::Vector{Float64}, y::Vector{Float64}, mu::Float64, residuals::F1,
                             nlconstraints::F2) where {F1<:Function, F2<:Function}
    rx, cx = residuals(x), nlconstraints(x)
    mx = 0.5*resid_sqnorm(rx) + dot(y,cx) + 0.5*mu*dot(cx,cx)     # :58
    return rx, cx, mx
end

# first_derivatives returns (y_bar, Jx, Cx, g) and the caller builds the AlHessian from Jx, Cx afterwards (:361-362).  The
# handle made here for g is keyed on ITS AlHessian object, so the caller's new object uploads J a second time; to avoid
# that, the AlHessian built here is remembered and handed to the next second_derivatives call with the same matrices.
const PENDING_HESS = Ref{Union{Nothing,BEnlsip.AlHessian{Float64}}}(nothing)

function BEnlsip.first_derivatives(x::Vector{Float64}, y::Vector{Float64}, mu::Float64, rx::Vector{Float64}, cx::Vector{Float64},
                                   jac_res::F1, jac_nlcons::F2) where {F1<:Function, F2<:Function}
    Jx, Cx = jac_res(x), jac_nlcons(x)
    y_bar = y + mu*cx
    H = BEnlsip.AlHessian(Jx,Cx,mu)
    g = gradient(H, rx, y_bar)                                     # :74
    PENDING_HESS[] = H
    return y_bar, Jx, Cx, g
end

function BEnlsip.second_derivatives(Jx::Matrix{Float64}, Cx::Matrix{Float64}, mu::Float64)
    H = PENDING_HESS[]
    PENDING_HESS[] = nothing
    (H !== nothing && H.J === Jx && H.C === Cx && H.mu == mu) && return H     # the object whose image is already in HBM
    return BEnlsip.AlHessian(Jx,Cx,mu)                                        # :84
end

function BEnlsip.least_squares_multipliers(x::Vector{Float64}, residuals::F1, jac_res::F2,
                                           jac_nlcons::F3) where {F1<:Function, F2<:Function, F3<:Function}
    Jx = jac_res(x)
    H = BEnlsip.AlHessian(Jx, zeros(0, size(Jx, 2)), 0.0)          # a throw-away image: J'r once per solve
    g = gradient(H, residuals(x), Float64[])                       # :893, summed over ranks
    finalize(HESS[H])
    delete!(HESS, H)
    C = jac_nlcons(x)
    chol_cct = cholesky(C*C')
    b = -C*g
    v = chol_cct.L \ b
    return chol_cct.U \ v
end

end # module
