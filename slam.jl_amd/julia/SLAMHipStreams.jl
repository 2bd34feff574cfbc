# SLAMHipStreams.jl -- the throughput path of libslamhip.so for a Julia host: S lock-stepped camera streams whose pyramids are
# built in one launch set per step and whose keypoint lists never leave HBM (include/slamhip.h, "lock-stepped batches of
# streams" and "device-resident keypoint lists"; DESIGN.md 3.10; INTEGRATION.md has the per-frame call sequence).
#
# SLAMHip.jl rebinds the reference's own seams one call at a time (one stream, host arrays in and out).  This file is what a
# host that runs many sequences at once -- the configuration the bench line is quoted on -- binds instead: the same C entry
# points Python's slam.jl_amd.PyramidBatch / KeypointSet wrap.  Device memory for the frames comes from the HIP runtime
# directly (hipMalloc / hipMemcpyAsync on the context's stream): no AMDGPU.jl, no Julia GPU codegen.
#
#     include("SLAMHip.jl"); include("SLAMHipStreams.jl")
#     SLAMHip.activate!("/path/to/libslamhip.so")
#     using .SLAMHipStreams
#     left = PyramidBatch(H, W, 3, S); prev = PyramidBatch(H, W, 3, S)
#     frames = FrameRing(H, W, S)                       # S uint8 frames in HBM + a pinned staging buffer
#     ks = KeypointSet(S, 2048)
#     upload!(frames, images_u8)                        # Vector of S Matrix{UInt8} (column-major H x W, KITTI decode)
#     update!(left, frames)                             # slam_pyr_update_batch_u8_dev
#     flow_match!(ks, prev, left, params; prior = 1)    # klt_tracking! for every stream
#     ...
#
# NOTE: like SLAMHip.jl, written against the Julia manual and include/slamhip.h, never executed (no Julia in the build
# container).  Every ccall mirrors the prototype of the same name in include/slamhip.h, argument for argument, so that a maintainer
# can check it by inspection.
module SLAMHipStreams

import ..SLAMHip: LIB, ctx, check

export PyramidBatch, FrameRing, KeypointSet, upload!, update!, flow_match!, stereo_match!, remove!, detect!, keyframe!,
       triangulate!, triangulate_temporal!, compute_pose_5pt!, compute_pose!, counts, download, stream_params

const HIP = "libamdhip64"
hipcheck(e::Cint, what) = e == 0 || error("$what: HIP error $e")

# ---- S pyramids in one allocation (slamhip.h: slam_pyr_create_batch) ---------------------------------------------------------
mutable struct PyramidBatch
    handles::Vector{Ptr{Cvoid}}          # the S members: ordinary slam_pyr handles (member 1 stands for the batch in the match calls)
    H::Int; W::Int; levels::Int
    function PyramidBatch(H, W, levels, S)
        hs = fill(C_NULL, S)
        check(ccall((:slam_pyr_create_batch, LIB[]), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}), ctx(), H, W, levels, S, hs))
        b = new(hs, H, W, levels)
        finalizer(x -> foreach(h -> ccall((:slam_pyr_destroy, LIB[]), Cint, (Ptr{Cvoid},), h), x.handles), b)   # plain C calls: no locks, no task switches
        b
    end
end
Base.length(b::PyramidBatch) = length(b.handles)

# ---- S uint8 frames in HBM, filled from the host through a pinned buffer on the context's stream ------------------------------
mutable struct FrameRing
    dev::Ptr{UInt8}; pinned::Ptr{UInt8}; ptrs::Vector{Ptr{UInt8}}; H::Int; W::Int; S::Int
    function FrameRing(H, W, S)
        d = Ref{Ptr{Cvoid}}(C_NULL); p = Ref{Ptr{Cvoid}}(C_NULL)
        hipcheck(ccall((:hipMalloc, HIP), Cint, (Ref{Ptr{Cvoid}}, Csize_t), d, H * W * S), "hipMalloc")
        hipcheck(ccall((:hipHostMalloc, HIP), Cint, (Ref{Ptr{Cvoid}}, Csize_t, Cuint), p, H * W * S, 0), "hipHostMalloc")
        f = new(Ptr{UInt8}(d[]), Ptr{UInt8}(p[]), [Ptr{UInt8}(d[]) + (s - 1) * H * W for s in 1:S], H, W, S)
        finalizer(x -> (ccall((:hipFree, HIP), Cint, (Ptr{Cvoid},), x.dev); ccall((:hipHostFree, HIP), Cint, (Ptr{Cvoid},), x.pinned)), f)
        f
    end
end

"frames: S matrices of UInt8 (H x W, column-major as Julia stores them).  Asynchronous on the context's stream; the pinned buffer
is reused by the next call, which therefore waits for this copy (slam_ctx_synchronize) first."
function upload!(f::FrameRing, frames::AbstractVector{<:AbstractMatrix{UInt8}})
    length(frames) == f.S || error("expected $(f.S) frames")
    check(ccall((:slam_ctx_synchronize, LIB[]), Cint, (Ptr{Cvoid},), ctx()))
    n = f.H * f.W
    for (s, img) in enumerate(frames)
        size(img) == (f.H, f.W) || error("frame $s has size $(size(img))")
        GC.@preserve img unsafe_copyto!(f.pinned + (s - 1) * n, pointer(img), n)
    end
    stream = ccall((:slam_ctx_stream, LIB[]), Ptr{Cvoid}, (Ptr{Cvoid},), ctx())
    hipcheck(ccall((:hipMemcpyAsync, HIP), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint, Ptr{Cvoid}), f.dev, f.pinned, n * f.S, 1, stream), "hipMemcpyAsync")   # 1 = hipMemcpyHostToDevice
    f
end

"update! of all S pyramids from the uint8 frames (raw / 255 on the device; slamhip.h: slam_pyr_update_batch_u8_dev).  target_only:
the batch is only ever matched INTO (right pyramids, mapper.jl:51-66) -- SLAM_PYR_TARGET_ONLY = 16.  tolerance = true selects mode 3:
with S >= 4 the tolerance-mode batch build (every plane within 1e-11 of the bit-exact one relative to the plane's magnitude, 2.2x instead
of 3.9x the algorithmic bytes; keypoint indices unaffected -- detect! works on the base layer); the default is the bit-exact mode 1."
function update!(b::PyramidBatch, f::FrameRing; σ = 1.0, target_only::Bool = false, tolerance::Bool = false, sync::Bool = false)
    hs = b.handles; ps = f.ptrs
    mode = (tolerance ? 3 : 1) | (target_only ? 16 : 0)
    GC.@preserve hs ps check(ccall((:slam_pyr_update_batch_u8_dev, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{UInt8}}, Cint, Cint, Cdouble, Cint),
        ctx(), hs, ps, length(hs), mode, Float64(σ), sync ? 1 : 0))
    b
end

# ---- per-stream call parameters (slamhip.h: S x 32 doubles) ---------------------------------------------------------------------
"params[:, s] = [Tcw (column-major 4 x 4) ; fx fy cx cy ; k1 k2 p1 p2 ; shift_y shift_x ; 0 ...]"
function stream_params(S; Tcw = nothing, cam = (1.0, 1.0, 0.0, 0.0), dist = (0.0, 0.0, 0.0, 0.0), shift = (0.0, 0.0))
    p = zeros(Float64, 32, S)
    for s in 1:S
        T = Tcw ≡ nothing ? [1.0 0 0 0; 0 1 0 0; 0 0 1 0; 0 0 0 1] : (Tcw isa AbstractVector ? Tcw[s] : Tcw)
        p[1:16, s] .= vec(Matrix{Float64}(T)); p[17:20, s] .= cam; p[21:24, s] .= dist; p[25:26, s] .= shift
    end
    p
end

# ---- the keypoint lists of S streams in HBM (slamhip.h: slam_kpset_*) -----------------------------------------------------------
mutable struct KeypointSet
    h::Ptr{Cvoid}; S::Int; cap::Int
    function KeypointSet(S, cap)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:slam_kpset_create, LIB[]), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}), ctx(), S, cap, r))
        k = new(r[], S, cap)
        finalizer(x -> ccall((:slam_kpset_destroy, LIB[]), Cint, (Ptr{Cvoid},), x.h), k)
        k
    end
end

"replace stream s's list (1-based s): yx 2 x n (y, x), is_3d n, xyz 3 x n"
function upload!(k::KeypointSet, s, yx::Matrix{Float64}, is_3d::Vector{UInt8}, xyz::Matrix{Float64})
    n = size(yx, 2)
    GC.@preserve yx is_3d xyz check(ccall((:slam_kpset_upload, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int64}, Cint),
        ctx(), k.h, s - 1, yx, is_3d, xyz, C_NULL, n))
    k
end

function download(k::KeypointSet, s)
    yx = zeros(2, k.cap); is3 = zeros(UInt8, k.cap); xyz = zeros(3, k.cap); ids = zeros(Int64, k.cap)
    syx = zeros(2, k.cap); hs = zeros(UInt8, k.cap); n = Ref{Cint}(0)
    check(ccall((:slam_kpset_download, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{UInt8}, Cint, Ref{Cint}),
        ctx(), k.h, s - 1, yx, is3, xyz, ids, syx, hs, k.cap, n))
    m = Int(n[])
    (yx = yx[:, 1:m], is_3d = is3[1:m] .!= 0, xyz = xyz[:, 1:m], ids = ids[1:m], stereo_yx = syx[:, 1:m], has_stereo = hs[1:m] .!= 0)
end

"the S list lengths: the one small device -> host copy of a step that ends without compute_pose!"
function counts(k::KeypointSet)
    c = zeros(Int32, k.S)
    check(ccall((:slam_kpset_counts, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}), ctx(), k.h, c))
    c
end

# optical_flow_matching!(map_manager, frame, from, to, false) for every stream (map_manager.jl:451-564)
function flow_match!(k::KeypointSet, from::PyramidBatch, to::PyramidBatch, params::Matrix{Float64}; prior = 0, pyramid_levels = 3,
                     pyramid_levels_3d = 1, window = 9, iterations = 30, eig_thr = 1e-4, eps = 0.01, max_distance = 1.0, n_bound = 0)   # max_distance = params.max_ktl_distance
    GC.@preserve params check(ccall((:slam_kpset_flow_match, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Cint, Cint, Cint, Cdouble, Cdouble, Cdouble, Cint),
        ctx(), k.h, from.handles[1], to.handles[1], params, prior, pyramid_levels, pyramid_levels_3d, window, iterations,
        eig_thr, eps, max_distance, n_bound))
    k
end

# optical_flow_matching!(..., true) + maybe_stereo_update! (map_manager.jl:579-590); params: the RIGHT camera
function stereo_match!(k::KeypointSet, left::PyramidBatch, right::PyramidBatch, params::Matrix{Float64}; prior = 2, pyramid_levels = 3,
                       pyramid_levels_3d = 1, window = 9, iterations = 30, eig_thr = 1e-4, eps = 0.01, max_distance = 1.0,
                       epipolar_error = 2.0, n_bound = 0)
    GC.@preserve params check(ccall((:slam_kpset_stereo_match, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Cint, Cint, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cint),
        ctx(), k.h, left.handles[1], right.handles[1], params, prior, pyramid_levels, pyramid_levels_3d, window, iterations,
        eig_thr, eps, max_distance, epipolar_error, n_bound))
    k
end

"flags_dev: S x cap bytes in HBM (slot order), e.g. written by the host's own culling kernel or uploaded with hipMemcpy"
remove!(k::KeypointSet, flags_dev::Ptr{UInt8}) = (check(ccall((:slam_kpset_remove, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}), ctx(), k.h, flags_dev)); k)

# extract_keypoints! (map_manager.jl:98-113): e is the reference's Extractor (max_points, radius, grid_resolution, cell_size)
function detect!(k::KeypointSet, cur::PyramidBatch, e; σ_mask = 3.0, min_response = 1e-4)
    check(ccall((:slam_kpset_detect, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Cint, Cdouble, Cdouble),
        ctx(), k.h, cur.handles[1], e.max_points, e.radius, e.grid_resolution[1], e.grid_resolution[2], e.cell_size, Float64(σ_mask), min_response))
    k
end

# create_keyframe! as far as the lists go (map_manager.jl:60-96); call after detect!
keyframe!(k::KeypointSet) = (check(ccall((:slam_kpset_keyframe, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), k.h)); k)

# triangulate_stereo! (mapper.jl:142-183): P1, P2 4 x 4 projection matrices, T21 right <- left, cam1 / cam2 = (fx, fy, cx, cy), Twc 16 x S
function triangulate!(k::KeypointSet, P1::Matrix{Float64}, P2::Matrix{Float64}, T21::Matrix{Float64}, cam1, cam2, Twc::Matrix{Float64};
                      max_error = 3.0, min_depth = 0.1, n_bound = 0)
    c1 = collect(Float64, cam1); c2 = collect(Float64, cam2)
    GC.@preserve P1 P2 T21 c1 c2 Twc check(ccall((:slam_kpset_triangulate, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cint),
        ctx(), k.h, P1, P2, T21, c1, c2, Twc, max_error, min_depth, n_bound))
    k
end

# triangulate_temporal! (mapper.jl:185-262).  kf_cw[s][q]: world -> camera of key-frame id q' with q' % nkf == q - 1 (a ring of the last
# nkf key-frame poses), Twc[s]: camera -> world of the frame, K4: the 4 x 4 calibration matrix; the four matrices per observer are the
# ones mapper.jl:226-231 forms.
function triangulate_temporal!(k::KeypointSet, params::Matrix{Float64}, kf_cw, Twc, K4::Matrix{Float64}, kf_cur::Vector{Int32};
                               kf_lo = nothing, max_error = 3.0, min_depth = 0.1, min_parallax = 20.0, n_bound = 0)
    nkf = length(kf_cw[1])
    tab = zeros(Float64, 64, nkf, k.S)
    for s in 1:k.S, q in 1:nkf
        rel = kf_cw[s][q] * Twc[s]; rel_inv = inv(rel)
        tab[1:16, q, s] .= vec(K4 * rel_inv); tab[17:32, q, s] .= vec(rel_inv); tab[33:48, q, s] .= vec(rel); tab[49:64, q, s] .= vec(inv(kf_cw[s][q]))
    end
    lo = kf_lo ≡ nothing ? Int32.(max.(kf_cur .- (nkf - 1), 0)) : kf_lo
    GC.@preserve params tab kf_cur lo check(ccall((:slam_kpset_triangulate_temporal, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Int32}, Ptr{Int32}, Cdouble, Cdouble, Cdouble, Cint),
        ctx(), k.h, params, tab, nkf, kf_cur, lo, max_error, min_depth, min_parallax, n_bound))
    k
end

# compute_pose_5pt! (front_end.jl:242-332): params[1:9, s] = R_compensation (column-major 3 x 3), [17:24] camera + distortion.
# fetch = false: enqueue only (the epipolar filter acts on the lists; compute_pose! right behind it brings the step's copy).
function compute_pose_5pt!(k::KeypointSet, params::Matrix{Float64}; min_parallax = 5.0, max_repr_error = 3.0, iters = 128, seed = UInt64(0), fetch = true)
    if !fetch
        GC.@preserve params check(ccall((:slam_kpset_compute_pose_5pt, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cdouble, Cdouble, Cint, UInt64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}),
            ctx(), k.h, params, min_parallax, max_repr_error, iters, seed, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
        return nothing
    end
    P = zeros(Float64, 12, k.S); status = zeros(Int32, k.S); ninl = zeros(Int32, k.S); par = zeros(Float64, k.S); cnt = zeros(Int32, k.S)
    GC.@preserve params check(ccall((:slam_kpset_compute_pose_5pt, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cdouble, Cdouble, Cint, UInt64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}),
        ctx(), k.h, params, min_parallax, max_repr_error, iters, seed, P, status, ninl, par, cnt))
    (Rt = [reshape(P[:, s], 3, 4) for s in 1:k.S], status = status, n_inliers = ninl, parallax = par, counts = cnt)
end

# compute_pose! (front_end.jl:132-219): P3P RANSAC + pnp_bundle_adjustment on the lists; the step's device -> host copy
function compute_pose!(k::KeypointSet, params::Matrix{Float64}; threshold = 3.0, iters = 256, seed = UInt64(0), pnp_iters_fast = 5,
                       pnp_iterations = 10, depth_eps = 1e-6, repr_eps = threshold)
    poses = zeros(Float64, 16, k.S); status = zeros(Int32, k.S); ninl = zeros(Int32, k.S); cnt = zeros(Int32, k.S)
    GC.@preserve params check(ccall((:slam_kpset_compute_pose, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cdouble, Cint, UInt64, Cint, Cint, Cdouble, Cdouble, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        ctx(), k.h, params, threshold, iters, seed, pnp_iters_fast, pnp_iterations, depth_eps, repr_eps, poses, status, ninl, cnt))
    (Tcw = [reshape(poses[:, s], 4, 4) for s in 1:k.S], status = status, n_inliers = ninl, counts = cnt)
end

# bundle_adjustment! (bundle_adjustment.jl:1-111) for the LocalBACaches of S lock-stepped SlamManagers in one set of launches
# (slam_local_ba_batch): what the S estimator tasks (estimator.jl:78-99, local_bundle_adjustment! :317-347) would each call once per
# key-frame.  caches: any objects with the LocalBACache fields (θ, θconst, pixels, poses_ids, points_ids, outliers, poses_remap,
# points_remap, observations -- estimator.jl:16-40); cameras: one Camera per cache.  Mutates every cache.θ / cache.outliers like
# S bundle_adjustment! calls; returns the per-window status codes (0 = solved; a window whose reduced system was not positive definite
# is left unchanged and reports -5).
function bundle_adjustment_batch!(caches::AbstractVector, cameras::AbstractVector; iterations::Int = 10, repr_ϵ::Real = 5.0)
    S = length(caches)
    Pn = Int32[length(c.poses_remap) for c in caches]; Mn = Int32[length(c.points_remap) for c in caches]; On = Int32[length(c.observations) for c in caches]
    cams = Float64[]; for cam in cameras; append!(cams, (cam.fx, cam.fy, cam.cx, cam.cy)); end
    θ = reduce(vcat, [Vector{Float64}(c.θ) for c in caches]); tc = reduce(vcat, [Vector{UInt8}(c.θconst) for c in caches])
    px = reduce(vcat, [vec(Matrix{Float64}(c.pixels)) for c in caches])                  # 2 x O column-major = (y, x) pairs back to back
    pid = reduce(vcat, [Vector{Int64}(c.poses_ids) for c in caches]); lid = reduce(vcat, [Vector{Int64}(c.points_ids) for c in caches])
    outl = zeros(UInt8, max(sum(On), 1)); status = zeros(Int32, S)
    GC.@preserve cams Pn Mn On θ tc px pid lid outl status check(ccall((:slam_local_ba_batch, LIB[]), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{UInt8},
         Cint, Cint, Cdouble, Ptr{Float64}, Ptr{Int32}),
        ctx(), S, cams, Pn, Mn, On, θ, tc, px, pid, lid, outl, 5, iterations, Float64(repr_ϵ), C_NULL, status))
    to = 0; oo = 0
    for (z, c) in enumerate(caches)
        n = 6 * Pn[z] + 3 * Mn[z]
        copyto!(c.θ, 1, θ, to + 1, n); to += n
        for i in 1:On[z]; c.outliers[i] = outl[oo + i] != 0; end
        oo += On[z]
    end
    status
end

# The same in two halves (slam_local_ba_batch_begin / _end): `begin` packs the caches and hands the call to a thread of the library -- the
# estimator task goes on (e.g. packs the next key-frame's caches) --, `finish!` waits and writes θ / outliers back.  The packed arrays live in the
# returned job until then; `jctx` is a context of the job's own (one job per context).
struct BABatchJob; jctx::Ptr{Cvoid}; caches; Pn; Mn; On; cams; θ; tc; px; pid; lid; outl; status; end
function bundle_adjustment_batch_begin(jctx::Ptr{Cvoid}, caches::AbstractVector, cameras::AbstractVector; iterations::Int = 10, repr_ϵ::Real = 5.0)
    S = length(caches)
    Pn = Int32[length(c.poses_remap) for c in caches]; Mn = Int32[length(c.points_remap) for c in caches]; On = Int32[length(c.observations) for c in caches]
    cams = Float64[]; for cam in cameras; append!(cams, (cam.fx, cam.fy, cam.cx, cam.cy)); end
    θ = reduce(vcat, [Vector{Float64}(c.θ) for c in caches]); tc = reduce(vcat, [Vector{UInt8}(c.θconst) for c in caches])
    px = reduce(vcat, [vec(Matrix{Float64}(c.pixels)) for c in caches])
    pid = reduce(vcat, [Vector{Int64}(c.poses_ids) for c in caches]); lid = reduce(vcat, [Vector{Int64}(c.points_ids) for c in caches])
    outl = zeros(UInt8, max(sum(On), 1)); status = zeros(Int32, S)
    job = BABatchJob(jctx, caches, Pn, Mn, On, cams, θ, tc, px, pid, lid, outl, status)       # (the job keeps every array alive until finish!)
    check(ccall((:slam_local_ba_batch_begin, LIB[]), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{UInt8},
         Cint, Cint, Cdouble, Ptr{Float64}, Ptr{Int32}),
        jctx, S, cams, Pn, Mn, On, θ, tc, px, pid, lid, outl, 5, iterations, Float64(repr_ϵ), C_NULL, status))
    job
end
function finish!(job::BABatchJob)
    check(ccall((:slam_local_ba_batch_end, LIB[]), Cint, (Ptr{Cvoid},), job.jctx))
    to = 0; oo = 0
    for (z, c) in enumerate(job.caches)
        n = 6 * job.Pn[z] + 3 * job.Mn[z]
        copyto!(c.θ, 1, job.θ, to + 1, n); to += n
        for i in 1:job.On[z]; c.outliers[i] = job.outl[oo + i] != 0; end
        oo += job.On[z]
    end
    job.status
end

# ---- one live stream, one call per frame (slam_frontend_*, include/slamhip.h): what run!()'s front-end task does per frame
# (front_end.jl:58-113, :454-470) and, at a key-frame, create_keyframe! + the mapper's stereo step (map_manager.jl:98-113, mapper.jl:51-66, :142-183),
# enqueued by ONE ccall; the keypoint list stays in HBM.  FrontEndConfig mirrors slam_frontend_config field for field (16 Int32, 9 Float64).
mutable struct FrontEndConfig
    H::Int32; W::Int32; pyramid_levels::Int32; pyramid_levels_3d::Int32; window::Int32; iterations::Int32
    max_points::Int32; radius::Int32; grid_rows::Int32; grid_cols::Int32; cell_size::Int32; cap::Int32
    pyr_mode::Int32; lookahead::Int32; right_target_only::Int32; reserved::Int32
    eig_thr::Float64; eps::Float64; max_distance::Float64; sigma_mask::Float64; min_response::Float64
    epipolar_error::Float64; max_error::Float64; min_depth::Float64; pyr_sigma::Float64
end
function FrontEndConfig(params, e; H::Integer, W::Integer, tolerance::Bool = false, lookahead::Bool = false)
    cells = e.grid_resolution[1] * e.grid_resolution[2]
    FrontEndConfig(H, W, params.pyramid_levels, 1, params.window_size, 30, e.max_points, e.radius, e.grid_resolution[1], e.grid_resolution[2], e.cell_size,
                   e.max_points + cells + 64, tolerance ? 3 : 1, lookahead ? 1 : 0, 1, 0,
                   1e-4, 1e-2, params.max_ktl_distance, 3.0, 1e-4, 2.0, params.max_reprojection_error, 0.1, params.pyramid_sigma)
end
struct LiveFrontEnd; h::Ptr{Cvoid}; cfg::FrontEndConfig; end
function LiveFrontEnd(cfg::FrontEndConfig; device::Integer = 0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:slam_frontend_create, LIB[]), Cint, (Cint, Ref{FrontEndConfig}, Ref{Ptr{Cvoid}}), device, cfg, r))
    LiveFrontEnd(r[], cfg)
end
close!(fe::LiveFrontEnd) = ccall((:slam_frontend_destroy, LIB[]), Cint, (Ptr{Cvoid},), fe.h)
# left / right: Matrix{UInt8} as the KITTI reader decodes them (column-major H x W); right === nothing unless the frame is a key-frame.  params /
# stereo_params: 32 Float64 each (stream_params of one stream); tri: 72 Float64 (P1, P2, T21, cam1, cam2, Twc).  Returns (frame index or -1, list length).
function step!(fe::LiveFrontEnd, left::Matrix{UInt8}, right::Union{Nothing, Matrix{UInt8}}, params::Vector{Float64}; prior::Integer = 1,
               stereo_params::Union{Nothing, Vector{Float64}} = nothing, stereo_prior::Integer = 2, tri::Union{Nothing, Vector{Float64}} = nothing)
    frame = Ref{Int32}(-1); n = Ref{Int32}(0)
    rc = GC.@preserve left right params stereo_params tri ccall((:slam_frontend_step, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{UInt8}, Ref{Int32}, Ref{Int32}),
        fe.h, left, right === nothing ? Ptr{UInt8}(C_NULL) : pointer(right), params, prior,
        stereo_params === nothing ? Ptr{Float64}(C_NULL) : pointer(stereo_params), stereo_prior, tri === nothing ? Ptr{Float64}(C_NULL) : pointer(tri), C_NULL, frame, n)
    rc == 0 || error("slam_frontend_step ($rc): ", unsafe_string(ccall((:slam_frontend_last_error, LIB[]), Cstring, (Ptr{Cvoid},), fe.h)))
    Int(frame[]), Int(n[])
end

end # module
