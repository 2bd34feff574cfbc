# SLAMHip.jl -- thin `ccall` shim that rebinds SLAM.jl's hot-path seams to
# libslamhip.so (hand-written HIP for MI355X / gfx950; include/slamhip.h).
#
# Usage (see INTEGRATION.md):
#
#     using SLAM
#     include("SLAMHip.jl"); SLAMHip.activate!("/path/to/libslamhip.so")
#     # ... build SlamManager and run!() exactly as before ...
#
# The shim keeps SLAM.jl's own types: `Extractor`, `LKPyramid{G,C}` and the
# untyped BA `cache`.  Nobody outside src/optical_flow/ reads the pyramid planes
# (grep-verified: only pyramid.jl and lucas_kanade.jl touch .layers/.Iy/...), so
# the Julia-side pyramid keeps its (small, unused) host planes and the device
# pyramid lives in a side table keyed by the identity of `lk.layers`.  No AMDGPU.jl, no Julia GPU
# codegen: every call below is a plain C call with host pointers.
#
# NOTE: Julia is not installed in the build container, so this file has been
# written against the Julia 1.6 manual and SLAM.jl's source but never executed.
module SLAMHip

using SLAM
using SLAM: Extractor, LKPyramid, LKCache, Point2f, Camera
using StaticArrays
using Images: Gray
using Random: randperm

const LIB = Ref{String}("libslamhip.so")

# ---- context: one per Julia task (SURVEY 8b: up to three OS threads call concurrently)
const CTX_LOCK = ReentrantLock()
const CTXS = Dict{UInt, Ptr{Cvoid}}()

function ctx()
    key = objectid(current_task())
    lock(CTX_LOCK) do
        get!(CTXS, key) do
            h = Ref{Ptr{Cvoid}}(C_NULL)
            rc = ccall((:slam_ctx_create, LIB[]), Cint, (Cint, Ref{Ptr{Cvoid}}), 0, h)
            rc == 0 || error("slam_ctx_create failed: ", unsafe_string(ccall((:slam_last_error, LIB[]), Cstring, (Ptr{Cvoid},), C_NULL)))
            h[]
        end
    end
end

@inline function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:slam_last_error, LIB[]), Cstring, (Ptr{Cvoid},), ctx()))
    rc == -3 ? throw(msg) : error("libslamhip ($rc): $msg")    # -3: "Not enough layers in pyramids." is `throw(String)` in the reference
end

# `Matrix{Gray{Float64}}` and `Matrix{Float64}` share their memory layout
@inline rawptr(image::Matrix{Gray{Float64}}) = Ptr{Float64}(pointer(image))
@inline rawptr(image::Matrix{Float64}) = pointer(image)

# ---- detect / describe (src/extractor.jl:63-105) -------------------------------
function hip_detect(e::Extractor, image, current_points; σ_mask = 3)
    length(current_points) ≥ e.max_points && return CartesianIndex{2}[]
    H, W = size(image)
    n_cells = e.grid_resolution[1] * e.grid_resolution[2]
    cap = n_cells * max(1, cld(e.max_points - length(current_points), n_cells))
    out = Matrix{Int64}(undef, 2, cap)
    n = Ref{Cint}(0)
    cur = reinterpret(Float64, current_points)            # Vector{SVector{2,Float64}} -> flat (y,x) pairs
    GC.@preserve image cur out check(ccall((:slam_detect, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Cint, Cint, Cint, Cint, Cint, Cint, Cdouble, Cdouble, Ptr{Int64}, Cint, Ref{Cint}),
        ctx(), rawptr(image), H, W, pointer(cur), length(current_points), e.max_points, e.radius,
        e.grid_resolution[1], e.grid_resolution[2], e.cell_size, Float64(σ_mask), 1e-4, out, cap, n))
    [CartesianIndex{2}(out[1, i], out[2, i]) for i in 1:n[]]
end

# ImageFeatures draws the BRIEF sampling pattern from a seeded Julia RNG; pass
# exactly that table down so descriptors match the pure-Julia path bit for bit.
function brief_table(e::Extractor)
    d = e.descriptor
    s1, s2 = d.sampling_type(d.size, d.window, d.seed)
    t = Matrix{Int32}(undef, 4, d.size)
    for k in 1:d.size
        t[1, k] = s1[k][1]; t[2, k] = s1[k][2]; t[3, k] = s2[k][1]; t[4, k] = s2[k][2]
    end
    t
end

function hip_describe(e::Extractor, image, keypoints)
    H, W = size(image)
    n = length(keypoints)
    d = e.descriptor
    rc = Matrix{Int64}(undef, 2, n)
    for (i, k) in enumerate(keypoints); rc[1, i] = k[1]; rc[2, i] = k[2]; end
    table = brief_table(e)
    words = d.size ÷ 64
    bits = Matrix{UInt64}(undef, words, n); orc = Matrix{Int64}(undef, 2, n); m = Ref{Cint}(0)
    GC.@preserve image rc table bits orc check(ccall((:slam_describe, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Int64}, Cint, Ptr{Int32}, Cint, Cdouble, Cint, Ptr{UInt64}, Ptr{Int64}, Ref{Cint}),
        ctx(), rawptr(image), H, W, rc, n, table, d.size, Float64(d.sigma), d.window, bits, orc, m))
    descriptors = BitVector[]
    for i in 1:m[]
        b = falses(d.size)
        for k in 0:(d.size - 1); b[k + 1] = (bits[k ÷ 64 + 1, i] >> (k % 64)) & 1 == 1; end
        push!(descriptors, b)
    end
    descriptors, [CartesianIndex{2}(orc[1, i], orc[2, i]) for i in 1:m[]]
end

# ---- LKPyramid (src/optical_flow/pyramid.jl) ------------------------------------
# Device twins.  `LKPyramid` is an immutable struct, so the device pyramid is tied to the IDENTITY of its (mutable)
# `layers` vector: side table objectid(lk.layers) => (WeakRef(lk.layers), handle) -- the WeakRef guards against a
# recycled objectid -- and a finalizer on the layers vector.  Finalizers must not block or switch tasks, so the finalizer
# (a closure over the handle and the key only, never over `lk`) just pushes the pair onto a queue under a try-locked spin
# lock, re-registering itself when the lock is busy; the queue is drained -- slam_pyr_destroy called, table entry
# removed -- from ordinary code (`register!`).
const PYR_LOCK = ReentrantLock()                    # ordinary code only, never taken in a finalizer
const PYRS = Dict{UInt, Tuple{WeakRef, Ptr{Cvoid}}}()
const DEAD = Tuple{UInt, Ptr{Cvoid}}[]
const DEAD_LOCK = Threads.SpinLock()

function release(layers, key::UInt, h::Ptr{Cvoid})
    if trylock(DEAD_LOCK)
        try
            push!(DEAD, (key, h))
        finally
            unlock(DEAD_LOCK)
        end
    else
        finalizer(l -> release(l, key, h), layers)  # lock busy (possibly on this very thread): try again at a later GC
    end
    nothing
end

function drain_dead()
    isempty(DEAD) && return
    dead = Tuple{UInt, Ptr{Cvoid}}[]
    GC.enable_finalizers(false)                     # `release` must not run on this thread while it holds the spin lock
    lock(DEAD_LOCK)
    try
        append!(dead, DEAD); empty!(DEAD)
    finally
        unlock(DEAD_LOCK)
        GC.enable_finalizers(true)
    end
    lock(PYR_LOCK) do
        for (key, h) in dead
            e = get(PYRS, key, nothing)
            e ≢ nothing && e[2] == h && delete!(PYRS, key)
            ccall((:slam_pyr_destroy, LIB[]), Cint, (Ptr{Cvoid},), h)
        end
    end
end

function register!(lk::LKPyramid, h::Ptr{Cvoid})
    drain_dead()
    layers = lk.layers
    key = objectid(layers)
    lock(() -> (PYRS[key] = (WeakRef(layers), h)), PYR_LOCK)
    finalizer(l -> release(l, key, h), layers)
    h
end

function handle(lk::LKPyramid; create_shape = nothing)
    e = lock(() -> get(PYRS, objectid(lk.layers), nothing), PYR_LOCK)
    e ≢ nothing && e[1].value ≡ lk.layers && return e[2]
    H, W = create_shape ≡ nothing ? size(lk.layers[1]) : create_shape
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:slam_pyr_create, LIB[]), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), ctx(), H, W, length(lk.layers) - 1, r))
    register!(lk, r[])
end

# LKPyramid(image, levels; σ, reusable): keep the Julia constructor for the host
# struct (its planes are never read again) and build the device twin with
# constructor semantics (mode 0).
function hip_pyramid(image, levels; σ = 1.0, reusable = true)
    # the generic (untyped) reference constructor stays reachable through `invoke`;
    # activate!() only adds a more specific method for Matrix{Gray{Float64}} images
    lk = invoke(SLAM.LKPyramid, Tuple{Any, Any}, image, levels; σ, reusable)   # host struct (frames 1-2 only: front_end.jl:459-467)
    GC.@preserve image check(ccall((:slam_pyr_update, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Cdouble),
        ctx(), handle(lk), rawptr(image), 0, Float64(σ)))
    lk
end

# target_only = true (SLAM_PYR_TARGET_ONLY): for a pyramid that is only ever matched INTO -- mapper.right_pyramid, whose one use is
# optical_flow_matching!(…, kf.left_pyramid, mapper.right_pyramid, true) (mapper.jl:51-66): levels above the finest get their layers
# only.  The generic update! method below cannot know the caller; mapper.jl:52 may call hip_update!(…; target_only = true) itself.
function hip_update!(lk::LKPyramid, img; σ = 1.0, target_only::Bool = false)
    GC.@preserve img check(ccall((:slam_pyr_update, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Cdouble),
        ctx(), handle(lk), rawptr(img), target_only ? (1 | 16) : 1, Float64(σ)))
    lk
end

function hip_copy!(dst::LKPyramid, src::LKPyramid)
    check(ccall((:slam_pyr_copy, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), ctx(), handle(dst), handle(src)))
    dst
end

function hip_deepcopy(lk::LKPyramid)                          # SLAM.jl:218: KF snapshot for the mapper task
    new = LKPyramid(map(copy, lk.layers), lk.Iy, lk.Ix, lk.Iyy, lk.Ixx, lk.Iyx, lk.cache)     # fresh `layers` vector = fresh identity
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:slam_pyr_clone, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx(), handle(lk), r))
    register!(new, r[])                                        # the clone is released with its layers vector like every other device pyramid
    new
end

# ---- fb_tracking! (src/tracker.jl:70-82) ----------------------------------------
function hip_fb_tracking!(previous_pyramid::LKPyramid, current_pyramid::LKPyramid, keypoints::AbstractVector{Point2f};
        displacement::Union{Nothing, AbstractVector{Point2f}} = nothing,
        iterations::Int = 30, window_size::Int = 11, pyramid_levels::Int = 3, max_distance::Real = 0.5)
    isempty(keypoints) && return
    n = length(keypoints)
    pts = collect(reinterpret(Float64, collect(keypoints)))
    d0buf = displacement ≡ nothing ? Float64[] : collect(reinterpret(Float64, collect(displacement)))   # rooted below
    out = Vector{Point2f}(undef, n); st = Vector{UInt8}(undef, n)
    GC.@preserve pts d0buf out st check(ccall((:slam_fb_track, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Cint, Cint, Cdouble, Cdouble, Cdouble, Ptr{Float64}, Ptr{UInt8}),
        ctx(), handle(previous_pyramid), handle(current_pyramid), pts,
        displacement ≡ nothing ? Ptr{Float64}(C_NULL) : pointer(d0buf), n, pyramid_levels, window_size, iterations,
        1e-4, 1e-2, Float64(max_distance), Ptr{Float64}(pointer(out)), st))
    out, BitVector(st .!= 0)
end

# ---- bundle_adjustment! / pnp_bundle_adjustment (src/bundle_adjustment.jl) -----
function hip_bundle_adjustment!(cache, camera::Camera; iterations::Int = 10, show_trace::Bool = false, repr_ϵ::Real = 5.0)
    P, M, O = length(cache.poses_remap), length(cache.points_remap), length(cache.observations)
    outl = Vector{UInt8}(undef, max(O, 1)); tc = Vector{UInt8}(cache.θconst)
    GC.@preserve cache outl tc check(ccall((:slam_local_ba, LIB[]), Cint,
        (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{UInt8}, Cint, Cint, Cdouble, Ptr{Float64}),
        ctx(), camera.fx, camera.fy, camera.cx, camera.cy, P, M, O, cache.θ, tc, cache.pixels, cache.poses_ids, cache.points_ids,
        outl, 5, iterations, Float64(repr_ϵ), C_NULL))
    for i in 1:O; cache.outliers[i] = outl[i] != 0; end
    cache.θ
end

function hip_pnp_bundle_adjustment(camera::Camera, pose::SMatrix{4, 4, Float64}, pixels, points;
        iterations::Int = 10, show_trace::Bool = false, depth_ϵ::Real = 1e-6, repr_ϵ::Real = 5.0)
    n = length(points)
    px = collect(reinterpret(Float64, collect(pixels))); pts = collect(reinterpret(Float64, collect(points)))
    pin = Vector{Float64}(vec(pose)); pout = Vector{Float64}(undef, 16)
    e0 = Ref{Cdouble}(0); e1 = Ref{Cdouble}(0); no = Ref{Cint}(0); outl = Vector{UInt8}(undef, max(n, 1))
    GC.@preserve px pts pin pout outl check(ccall((:slam_pnp_ba, LIB[]), Cint,
        (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Cint, Cint, Cdouble, Cdouble,
         Ptr{Float64}, Ref{Cdouble}, Ref{Cdouble}, Ptr{UInt8}, Ref{Cint}),
        ctx(), camera.fx, camera.fy, camera.cx, camera.cy, pin, px, pts, n, 5, iterations, Float64(depth_ϵ), Float64(repr_ϵ),
        pout, e0, e1, outl, no))
    SMatrix{4, 4, Float64}(pout), e0[], e1[], Bool[o != 0 for o in outl[1:n]], Int(no[])
end

# Array-level body of triangulate_stereo! / triangulate_temporal! (mapper.jl:142-262): the per-keypoint
# `triangulate` + gates for a whole keypoint list in one launch.  Not bound by activate!: the reference interleaves
# map surgery with the per-keypoint arithmetic, so a maintainer calls this once per key-frame from a vectorised
# triangulate_stereo! (collect the candidate keypoints, call, then apply update_mappoint! / remove_* from `status`).
# `parallax = nothing`: stereo semantics; otherwise per-keypoint parallax values (temporal semantics).
function hip_triangulate(cam1::Camera, cam2::Camera, P1::SMatrix{4, 4, Float64}, P2::SMatrix{4, 4, Float64}, T21::SMatrix{4, 4, Float64},
        px1, px2, max_error; min_depth = 0.1, parallax = nothing, min_parallax = 20.0)
    n = length(px1)
    a = collect(reinterpret(Float64, collect(px1))); b = collect(reinterpret(Float64, collect(px2)))   # (y, x) pairs
    p1 = Vector{Float64}(vec(P1)); p2 = Vector{Float64}(vec(P2)); t = Vector{Float64}(vec(T21))
    c1 = Float64[cam1.fx, cam1.fy, cam1.cx, cam1.cy]; c2 = Float64[cam2.fx, cam2.fy, cam2.cx, cam2.cy]
    par = parallax === nothing ? C_NULL : pointer(parallax)
    out = Vector{Float64}(undef, 3 * max(n, 1)); st = Vector{UInt8}(undef, max(n, 1))
    GC.@preserve a b p1 p2 t c1 c2 parallax out st check(ccall((:slam_triangulate, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint,
         Cdouble, Cdouble, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{UInt8}),
        ctx(), p1, p2, t, c1, c2, a, b, n, Float64(max_error), Float64(min_depth), par, Float64(min_parallax), out, st))
    [SVector{3, Float64}(out[3i - 2], out[3i - 1], out[3i]) for i in 1:n], Bool[s != 0 for s in st[1:n]]
end

# p3p_ransac of compute_pose! (front_end.jl:164-167): same positional arguments and result shape as
# RecoverPose.p3p_ransac -- `(n_inliers, (KP, inliers, error))`, or `nothing` when no sample gave a pose.  The
# sample triples are drawn here (Julia's RNG) and handed to the library 0-based; `iterations` triples are all scored.
function hip_p3p_ransac(points, pixels, pdn_positions, K; threshold = 1.0, iterations = 256)
    n = length(points)
    pts = collect(reinterpret(Float64, collect(SVector{3, Float64}.(points))))
    px = collect(reinterpret(Float64, collect(SVector{2, Float64}.(pixels))))       # already (x, y), front_end.jl:151
    pdn = collect(reinterpret(Float64, collect(SVector{3, Float64}.(pdn_positions))))
    k = Vector{Float64}(vec(SMatrix{3, 3, Float64}(K)))
    smp = Vector{Int32}(undef, 3 * iterations)
    for it in 0:iterations - 1
        smp[3it + 1:3it + 3] .= Int32.(randperm(n)[1:3] .- 1)
    end
    kp = Vector{Float64}(undef, 12); inl = Vector{UInt8}(undef, max(n, 1))
    cnt = Ref{Cint}(0); err = Ref{Cdouble}(0.0)
    GC.@preserve pts px pdn k smp kp inl check(ccall((:slam_p3p_ransac, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Cdouble, Ptr{Int32}, Cint,
         Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}, Ref{Cdouble}, Ptr{Cint}),
        ctx(), pts, px, pdn, n, k, Float64(threshold), smp, iterations, kp, C_NULL, inl, cnt, err, C_NULL))
    cnt[] == 0 && return nothing
    Int(cnt[]), (SMatrix{3, 4, Float64}(kp), Bool[i != 0 for i in inl[1:n]], err[])
end

# five_point_ransac of compute_pose_5pt! (front_end.jl:305-308): same positional arguments as
# RecoverPose.five_point_ransac (the GEEV cache is not needed) and the result shape the caller destructures,
# `n_inliers, (E, P, inliers, error)`; P = [R | t] (3x4, previous key-frame -> current frame, |t| = 1).
function hip_five_point_ransac(previous_points, current_points, previous_pd, current_pd, K1, K2, cache = nothing;
        max_repr_error = 1.0, iterations = 128)
    n = length(previous_points)
    flat(v) = collect(reinterpret(Float64, collect(SVector{2, Float64}.(v))))
    a = flat(previous_points); b = flat(current_points); c = flat(previous_pd); d = flat(current_pd)   # already (x, y)
    k1 = Vector{Float64}(vec(SMatrix{3, 3, Float64}(K1))); k2 = Vector{Float64}(vec(SMatrix{3, 3, Float64}(K2)))
    smp = Vector{Int32}(undef, 5 * iterations)
    for it in 0:iterations - 1
        smp[5it + 1:5it + 5] .= Int32.(randperm(n)[1:5] .- 1)
    end
    E = Vector{Float64}(undef, 9); P = Vector{Float64}(undef, 12); inl = Vector{UInt8}(undef, max(n, 1))
    cnt = Ref{Cint}(0); err = Ref{Cdouble}(0.0)
    GC.@preserve a b c d k1 k2 smp E P inl check(ccall((:slam_five_point_ransac, LIB[]), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Cdouble,
         Ptr{Int32}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}, Ref{Cdouble}, Ptr{Cint}),
        ctx(), a, b, c, d, n, k1, k2, Float64(max_repr_error), smp, iterations, E, P, inl, cnt, err, C_NULL))
    Int(cnt[]), (SMatrix{3, 3, Float64}(E), SMatrix{3, 4, Float64}(P), Bool[i != 0 for i in inl[1:n]], err[])
end

"""
    activate!(libpath)

Rebind SLAM.jl's seams to libslamhip.  Method (re)definition on the original
generic functions: later calls from front_end.jl / map_manager.jl / mapper.jl /
estimator.jl dispatch here without any change to those files.
"""
function activate!(libpath::AbstractString = "libslamhip.so")
    LIB[] = libpath
    @eval SLAM begin
        detect(e::Extractor, image, current_points; σ_mask = 3) = $(hip_detect)(e, image, current_points; σ_mask)
        describe(e::Extractor, image, keypoints) = $(hip_describe)(e, image, keypoints)
        update!(lk::LKPyramid{G, C}, img; σ = 1.0) where {G <: AbstractVector, C} = $(hip_update!)(lk, img; σ)
        Base.copy!(dst::LKPyramid{G, C}, src::LKPyramid{G, C}) where {G, C} = $(hip_copy!)(dst, src)
        Base.deepcopy_internal(lk::LKPyramid, ::IdDict) = $(hip_deepcopy)(lk)
        fb_tracking!(p::LKPyramid, c::LKPyramid, k::AbstractVector{Point2f}; kwargs...) = $(hip_fb_tracking!)(p, c, k; kwargs...)
        bundle_adjustment!(cache, camera::Camera; kwargs...) = $(hip_bundle_adjustment!)(cache, camera; kwargs...)
        pnp_bundle_adjustment(camera::Camera, pose::SMatrix{4, 4, Float64}, pixels, points; kwargs...) =
            $(hip_pnp_bundle_adjustment)(camera, pose, pixels, points; kwargs...)
    end
    # the two constructor call sites (front_end.jl:464, mapper.jl:54) call LKPyramid(image, levels; ...):
    @eval SLAM LKPyramid(image::Matrix{Gray{Float64}}, levels::Int; σ = 1.0, reusable = false, kwargs...) =
        $(hip_pyramid)(image, levels; σ, reusable = true)
    nothing
end

end # module
