# make_golden_julia.jl -- reference-side fixture generator (run by a maintainer who has Julia + SLAM.jl).
#
# Julia is not installed in the build container and SLAM.jl ships no golden vectors, so the CPU oracle in oracle/ is
# "parity unpinned".  This script closes that gap from the reference side: it feeds the INPUTS of
# tests/golden/hotpath_v1.npz to the real SLAM.jl functions on the hot path and writes their outputs to
# tests/golden/julia_v1.npz under the same key names.  tests/test_golden_julia.py picks the file up when it exists and
# compares the oracle (CPU) and the HIP path (GPU) against it -- that is what pins the oracle.
#
#     cd /path/to/SLAM.jl
#     julia --project=. -e 'using Pkg; Pkg.add("NPZ")'
#     julia --project=. /path/to/repo/tests/golden/make_golden_julia.jl /path/to/repo/tests/golden
#
# Besides the seam outputs it records the upstream-package primitives the oracle restates from their published
# semantics (SURVEY Appendix A) on the same image -- shi_tomasi of one cell, the IIR Gaussian, imresize, the avoidance
# mask, the Scharr gradients -- so that a mismatch at a seam can be localised to the primitive that differs.
using SLAM
using NPZ
using Images, ImageFiltering, ImageDraw
using StaticArrays
using LinearAlgebra

dir = length(ARGS) ≥ 1 ? ARGS[1] : @__DIR__
G = npzread(joinpath(dir, "hotpath_v1.npz"))

gray(u8) = Gray{Float64}.(Float64.(u8) ./ 255)           # example/kitty/main.jl:39-41
raw(m) = Float64.(m)
img0, img1 = gray(G["img0_u8"]), gray(G["img1_u8"])
H, W = size(img0)
out = Dict{String, Any}()

# ---- detect (src/extractor.jl:63-95), parameters as SLAM.jl:149-160 builds them with max_distance = 35
cell = 35
e = SLAM.Extractor(60, 17, (cld(H, cell), cld(W, cell)), cell)
pts(m) = [SLAM.Point2f(m[i, 1], m[i, 2]) for i in 1:size(m, 1)]
to_mat(v) = isempty(v) ? zeros(Int64, 0, 2) : permutedims(hcat([[k[1], k[2]] for k in v]...))
out["kp_nomask"] = to_mat(SLAM.detect(e, img0, SLAM.Point2f[]))
out["kp_mask"] = to_mat(SLAM.detect(e, img0, pts(G["cur"])))

# primitives behind detect
out["prim_mask"] = raw(SLAM.get_mask(img0, pts(G["cur"]), e.radius))
out["prim_mask_blurred"] = raw(imfilter(SLAM.get_mask(img0, pts(G["cur"]), e.radius), Kernel.gaussian(3)))
out["prim_shi_tomasi_cell11"] = raw(shi_tomasi(@view(img0[1:cell, 1:cell])))

# ---- LKPyramid ctor / update! (src/optical_flow/pyramid.jl:40-137), 2 levels above the base like the fixture
p0 = SLAM.LKPyramid(img0, 2; σ = 1.0, reusable = true); SLAM.update!(p0, img0)
p1 = SLAM.LKPyramid(img1, 2; σ = 1.0, reusable = true); SLAM.update!(p1, img1)
pc = SLAM.LKPyramid(img0, 2; σ = 1.0, reusable = true)
out["upd_Iy_l1"] = raw(p0.Iy[2]); out["upd_Iyx_l2"] = raw(p0.Iyx[3]); out["upd_layer_l2"] = raw(p0.layers[3])
out["ctor_Ixx_l1"] = raw(pc.Ixx[2]); out["ctor_layer_l1"] = raw(pc.layers[2])
for (name, planes) in (("layers", p0.layers), ("Iy", p0.Iy), ("Ix", p0.Ix), ("Iyy", p0.Iyy), ("Ixx", p0.Ixx), ("Iyx", p0.Iyx)), l in 1:3
    out["upd_full_$(name)_l$(l - 1)"] = raw(planes[l])
end
# primitives behind the pyramid
kern1 = KernelFactors.IIRGaussian(1.0)
out["prim_iir_sigma1_replicate"] = raw(imfilter(img0, (kern1, kern1), "replicate"))
out["prim_iir_sigma4_replicate"] = raw(imfilter(img0, (KernelFactors.IIRGaussian(4.0), KernelFactors.IIRGaussian(4.0)), "replicate"))
out["prim_iir_sigma1_NA"] = raw(imfilter(img0, (kern1, kern1), NA()))
out["prim_imresize_half"] = raw(imresize(img0, (cld(H, 2), cld(W, 2))))
gy, gx = imgradients(img0, KernelFactors.scharr, "replicate")
out["prim_scharr_y"] = raw(gy); out["prim_scharr_x"] = raw(gx)

# ---- fb_tracking! (src/tracker.jl:70-82) with the arguments optical_flow_matching! passes (map_manager.jl:549-552)
kps = [SLAM.Point2f(Float64(out["kp_nomask"][i, 1]), Float64(out["kp_nomask"][i, 2])) for i in 1:size(out["kp_nomask"], 1)]
res = SLAM.fb_tracking!(p0, p1, kps; pyramid_levels = 2, window_size = 9, max_distance = 1.0)
new_kps, status = res
lk_out = zeros(Float64, length(kps), 2)
for i in 1:length(kps)
    status[i] && (lk_out[i, :] .= new_kps[i])
end
out["lk_out"] = lk_out; out["lk_status"] = UInt8.(status)

# ---- bundle_adjustment! (src/bundle_adjustment.jl:1-55) on the fixture's flat arrays
cam = G["ba_cam"]
camera = SLAM.Camera(; fx = cam[1], fy = cam[2], cx = cam[3], cy = cam[4], height = 376, width = 1241)
θ = copy(G["ba_theta0"]); P = length(G["ba_const"]); O = length(G["ba_pose_ids"]); M = (length(θ) - 6P) ÷ 3
dummy = SLAM.Observation(SLAM.Point2f(0, 0), SLAM.Point3f(0, 0, 0), ntuple(_ -> 0.0, 6), 0, 0, false, false, 0, 0)
cache = SLAM.LocalBACache(fill(dummy, O), Set{Int64}(), θ, Bool.(G["ba_const"]), permutedims(G["ba_pixels"]),
                          Int64.(G["ba_pose_ids"]), Int64.(G["ba_point_ids"]), collect(1:P), collect(1:M))
SLAM.bundle_adjustment!(cache, camera; iterations = 10, repr_ϵ = 5.0)
out["ba_theta"] = copy(cache.θ); out["ba_outliers"] = UInt8.(cache.outliers)
# final cost at the returned parameters, outliers zeroed (the quantity LeastSquaresOptim minimises in pass 2)
function ssr(θ)
    s = 0.0
    for i in 1:O
        cache.outliers[i] && continue
        T = θ[(6 * (cache.poses_ids[i] - 1) + 1):(6 * cache.poses_ids[i])]
        X = θ[(6P + 3 * (cache.points_ids[i] - 1) + 1):(6P + 3 * cache.points_ids[i])]
        pt = SLAM.RotZYX(T[1:3]...) * SVector{3}(X) .+ SVector{3}(T[4:6])
        py = cam[2] * pt[2] / pt[3] + cam[4]; px = cam[1] * pt[1] / pt[3] + cam[3]
        s += (cache.pixels[1, i] - py)^2 + (cache.pixels[2, i] - px)^2
    end
    s
end
out["ba_ssr_final"] = [ssr(cache.θ)]

# ---- describe (src/extractor.jl:103-105): the BRIEF sampling table comes from Julia's seeded RNG -- record it
desc, kept = SLAM.describe(e, img0, [CartesianIndex(out["kp_nomask"][i, 1], out["kp_nomask"][i, 2]) for i in 1:size(out["kp_nomask"], 1)])
d = e.descriptor
s1, s2 = d.sampling_type(d.size, d.window, d.seed)
pat = zeros(Int32, d.size, 4)
for k in 1:d.size
    pat[k, 1] = s1[k][1]; pat[k, 2] = s1[k][2]; pat[k, 3] = s2[k][1]; pat[k, 4] = s2[k][2]
end
bits = zeros(UInt64, length(desc), d.size ÷ 64)
for (i, b) in enumerate(desc), k in 0:(d.size - 1)
    b[k + 1] && (bits[i, k ÷ 64 + 1] |= UInt64(1) << (k % 64))
end
out["brief_pattern"] = pat; out["brief_bits"] = bits; out["brief_rc"] = to_mat(kept)

# ---- ReplaySaver (src/io/saver.jl): three frames, one of them set twice; slam.jl_amd/saver.py mirrors it (BSON lowering unpinned)
saver = SLAM.ReplaySaver()
wc(t) = SMatrix{4, 4, Float64, 16}([1.0 0 0 t[1]; 0 1 0 t[2]; 0 0 1 t[3]; 0 0 0 1])
SLAM.set_frame_wc!(saver, 7, wc((1.0, 2.0, 3.0))); SLAM.set_frame_wc!(saver, 9, wc((-4.0, 0.5, 6.0)))
SLAM.set_frame_wc!(saver, 7, wc((1.5, 2.5, 3.5))); SLAM.set_frame_wc!(saver, 12, wc((0.0, 0.0, 10.0)))
SLAM.save(saver, joinpath(dir, "julia_replay"))

out["versions"] = [string(VERSION)]
npzwrite(joinpath(dir, "julia_v1.npz"), Dict(k => (v isa Vector{String} ? codeunits(join(v, ";")) |> collect : v) for (k, v) in out))
println("wrote ", joinpath(dir, "julia_v1.npz"), ": ", size(out["kp_nomask"], 1), " keypoints, ", sum(status), " tracked")
