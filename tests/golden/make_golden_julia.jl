# make_golden_julia.jl -- reference-side fixture generator (run by a maintainer who has Julia + SLAM.jl).
#
# Julia is not installed in the build container and SLAM.jl ships no golden vectors, so the CPU oracle in oracle/ is
# "parity unpinned".  This script closes that gap from the reference side: it feeds the INPUTS of
# tests/golden/hotpath_v1.npz to the real SLAM.jl functions on the hot path and writes their outputs to
# tests/golden/julia_v1.npz under the same key names.  tests/test_golden_julia.py picks the file up when it exists and
# compares the oracle (CPU) and the HIP path (GPU) against it -- that is what pins the oracle.
#
#     cd /path/to/SLAM.jl                                    # at the commit this repository was built against (Project.toml: Images 0.24,
#     julia --project=. -e 'using Pkg; Pkg.instantiate()'    #  ImageFiltering 0.6/0.7, ImageDraw 0.2, ImageFeatures 0.4, Interpolations 0.13,
#     julia --project=. -e 'using Pkg; Pkg.add(name="NPZ", version="0.4")'   #  LeastSquaresOptim 0.8, SparseDiffTools 1, Rotations 1, BSON 0.3, RecoverPose 0.1)
#     julia --project=. /path/to/repo/tests/golden/make_golden_julia.jl /path/to/repo/tests/golden
#
# One command writes every fixture the skipped tests wait for: julia_v1.npz (part 1), julia_frontend_v1.npz (part 2) and the
# directory julia_replay/ (ReplaySaver BSON files).  `Pkg.status()` of the run is stored in julia_v1.npz["versions"].
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
# (round 4) the primitives a first run needs to LOCALISE an Appendix-A error -- each is one upstream call on a small input:
#  * the constructor's gradient border (pyramid.jl:51,59: Fill(0)) beside update!'s replicate border above
gy0, gx0 = imgradients(img0, KernelFactors.scharr, Fill(zero(eltype(img0))))
out["prim_scharr_fill0_y"] = raw(gy0); out["prim_scharr_fill0_x"] = raw(gx0)
#  * shi_tomasi's 3 x 3 box mean ALONE (Images 0.24 corner.jl: imfilter(cxx, centered(ones(3, 3) ./ 9)) -- dense, or SVD-factored into two
#    passes by ImageFiltering: the oracle applies 1/3 (x) 1/3 separably; a 1-ulp difference here flips strict maxima and keypoint indices)
cell11 = @view(img0[1:cell, 1:cell])
sgx, sgy = imgradients(cell11, KernelFactors.sobel, "replicate")
out["prim_sobel_y_cell11"] = raw(sgy); out["prim_sobel_x_cell11"] = raw(sgx)
out["prim_box3_of_gy2_cell11"] = raw(imfilter(sgy .* sgy, centered(ones(3, 3) ./ 9)))
#  * findlocalmaxima of the cell response (strict 8-neighbour maxima, edges included, column-major order) and the stable descending order
resp11 = shi_tomasi(cell11)
mx = findlocalmaxima(resp11)
out["prim_localmaxima_cell11"] = isempty(mx) ? zeros(Int64, 0, 2) : permutedims(hcat([[k[1], k[2]] for k in mx]...))
ord = sortperm([resp11[k] for k in mx]; lt = (x, y) -> x > y)
out["prim_localmaxima_order_cell11"] = Int64.(ord)
#  * Images.boxdiff on an integral image (lucas_kanade.jl:143-145), windows touching the first row / column (the index-0 rule) and interior
ii = integral_image(raw(img0))
out["prim_boxdiff"] = Float64[boxdiff(ii, 1:5, 1:7), boxdiff(ii, 3:21, 1:19), boxdiff(ii, 1:19, 4:22), boxdiff(ii, 10:28, 15:33), boxdiff(ii, (H - 18):H, (W - 18):W)]

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

import Pkg
out["versions"] = [string(VERSION); ["$(p.name)=$(p.version)" for p in values(Pkg.dependencies()) if p.is_direct_dep]]
npzwrite(joinpath(dir, "julia_v1.npz"), Dict(k => (v isa Vector{String} ? codeunits(join(v, ";")) |> collect : v) for (k, v) in out))
println("wrote ", joinpath(dir, "julia_v1.npz"), ": ", size(out["kp_nomask"], 1), " keypoints, ", sum(status), " tracked")

# =====================================================================================================================
# Part 2 (round 3): the rows SURVEY 8f added -- optical_flow_matching! (src/map_manager.jl:451-564, temporal and stereo with
# maybe_stereo_update! :579-590), triangulate_stereo! (src/mapper.jl:142-183), RecoverPose's triangulate / p3p_ransac /
# five_point_ransac as compute_pose! (src/front_end.jl:164-167) and compute_pose_5pt! (:305-308) call them.  Inputs:
# tests/golden/frontend_v1.npz and tests/golden/pose_v1.npz; output: tests/golden/julia_frontend_v1.npz.
# =====================================================================================================================
using RecoverPose
F = npzread(joinpath(dir, "frontend_v1.npz"))
fo = Dict{String, Any}()
l0, l1, r1 = gray(F["l0_u8"]), gray(F["l1_u8"]), gray(F["r1_u8"])
FH, FW = size(l0)
fcam = F["cam"]; baseline = F["baseline"][1]
Ti0 = SMatrix{4, 4, Float64, 16}([1.0 0 0 -baseline; 0 1 0 0; 0 0 1 0; 0 0 0 1])      # right camera: x_right = x_left - baseline
cam_l = SLAM.Camera(; fx = fcam[1], fy = fcam[2], cx = fcam[3], cy = fcam[4], height = FH, width = FW)
cam_r = SLAM.Camera(; fx = fcam[1], fy = fcam[2], cx = fcam[3], cy = fcam[4], height = FH, width = FW, Ti0)
fparams = SLAM.Params(; stereo = true, pyramid_levels = 3, window_size = 9, max_ktl_distance = 1.0)
fex = SLAM.Extractor(80, 17, (cld(FH, 35), cld(FW, 35)), 35)
q0 = SLAM.LKPyramid(l0, 3; σ = 1.0, reusable = true); SLAM.update!(q0, l0)
q1 = SLAM.LKPyramid(l1, 3; σ = 1.0, reusable = true); SLAM.update!(q1, l1)
qr = SLAM.LKPyramid(r1, 3; σ = 1.0, reusable = true); SLAM.update!(qr, r1)

# a Frame + MapManager carrying exactly the fixture's keypoints: keypoint i has id i; a 3-D keypoint's map point sits where the
# frame (cw = I) projects it onto the fixture's prior `proj[i]` (depth 10), so project_world_to_image_distort returns proj[i]
function build_frame(kp, is3d, proj; depth = 10.0, right = false)
    frame = SLAM.Frame(; camera = cam_l, right_camera = cam_r, cell_size = 35, id = 1, kfid = 1)
    mm = SLAM.MapManager(fparams, frame, fex)
    for i in 1:size(kp, 1)
        SLAM.add_keypoint!(frame, SLAM.Point2f(kp[i, 1], kp[i, 2]), i; is_3d = Bool(is3d[i]))
        mp = SLAM.MapPoint(i, 1, BitVector(), true)
        if Bool(is3d[i])
            # right image: x_right = x_left - baseline  ->  the left-camera point whose RIGHT projection is proj[i]
            xs = (proj[i, 2] - fcam[3]) / fcam[1] * depth + (right ? baseline : 0.0)
            ys = (proj[i, 1] - fcam[4]) / fcam[2] * depth
            SLAM.set_position!(mp, SLAM.Point3f(xs, ys, depth))
        end
        mm.map_points[i] = mp
    end
    frame, mm
end
function dump_frame(frame, n)
    pix = fill(NaN, n, 2); present = zeros(UInt8, n); stereo = zeros(UInt8, n); rpix = fill(NaN, n, 2); is3 = zeros(UInt8, n)
    for (id, k) in frame.keypoints
        pix[id, :] .= k.pixel; present[id] = 1; is3[id] = k.is_3d
        if k.is_stereo
            stereo[id] = 1; rpix[id, :] .= k.right_pixel
        end
    end
    pix, present, stereo, rpix, is3
end

# ---- temporal: optical_flow_matching!(map_manager, frame, from, to, false)
kp, is3d, proj = F["kp"], F["is3d"], F["proj"]
n = size(kp, 1)
frame, mm = build_frame(kp, is3d, proj)
SLAM.optical_flow_matching!(mm, frame, q0, q1, false)
pix, present, _, _, _ = dump_frame(frame, n)
fo["t_new"] = pix; fo["t_present"] = present                       # present = 0: observation removed (:559)

# ---- stereo: the surviving keypoints at their tracked positions, matched into the right image; then triangulate_stereo!
keepi = findall(present .== 1)
kp1 = pix[keepi, :]; is3d1 = is3d[keepi]
sproj = F["s_proj"]
@assert size(sproj, 1) == length(keepi) "the oracle kept $(size(sproj, 1)) keypoints, Julia kept $(length(keepi)): the temporal match already differs"
frame2, mm2 = build_frame(kp1, is3d1, sproj; right = true)
SLAM.optical_flow_matching!(mm2, frame2, q1, qr, true)
pix2, present2, stereo2, rpix2, _ = dump_frame(frame2, length(keepi))
fo["s_present"] = present2; fo["s_stereo"] = stereo2; fo["s_right"] = rpix2
SLAM.triangulate_stereo!(mm2, frame2, 3.0, RecoverPose.GEEV4x4Cache())
tri3 = zeros(UInt8, length(keepi)); trixyz = fill(NaN, length(keepi), 3); tristereo = zeros(UInt8, length(keepi))
for (id, k) in frame2.keypoints
    tristereo[id] = k.is_stereo
    mp = get(mm2.map_points, id, nothing)
    if mp !== nothing && mp.is_3d
        tri3[id] = 1; trixyz[id, :] .= SLAM.get_position(mp)
    end
end
fo["tri_is3d"] = tri3; fo["tri_xyz"] = trixyz; fo["tri_stereo_after"] = tristereo

# ---- RecoverPose primitives on the inputs of pose_v1.npz
Pz = npzread(joinpath(dir, "pose_v1.npz"))
tc = Pz["tri_cam"]; T21 = SMatrix{4, 4, Float64, 16}(Pz["tri_T21"])
K4(c) = SMatrix{4, 4, Float64, 16}([c[1] 0 c[3] 0; 0 c[2] c[4] 0; 0 0 1 0; 0 0 0 1])
P1 = K4(tc) * SMatrix{4, 4, Float64, 16}(I); P2 = K4(tc) * T21
cache = RecoverPose.GEEV4x4Cache()
ntri = size(Pz["tri_px1"], 1)
tri = zeros(Float64, ntri, 4)
for i in 1:ntri                                                        # mapper.jl:162-165: (x, y) pixels
    tri[i, :] .= RecoverPose.triangulate(SLAM.Point2f(Pz["tri_px1"][i, 2], Pz["tri_px1"][i, 1]), SLAM.Point2f(Pz["tri_px2"][i, 2], Pz["tri_px2"][i, 1]), P1, P2, cache)
end
fo["rp_triangulate_h"] = tri                                          # homogeneous, before the division by [4]
K3(m) = SMatrix{3, 3, Float64, 9}(m)
p3pts = [SLAM.Point3f(Pz["p3p_pts"][i, :]...) for i in 1:size(Pz["p3p_pts"], 1)]
p3px = [SLAM.Point2f(Pz["p3p_px"][i, :]...) for i in 1:size(Pz["p3p_px"], 1)]
p3pdn = [SLAM.Point3f(Pz["p3p_pdn"][i, :]...) for i in 1:size(Pz["p3p_pdn"], 1)]
res = RecoverPose.p3p_ransac(p3pts, p3px, p3pdn, K3(Pz["p3p_K"]); threshold = 3.0)           # front_end.jl:164-167
if res !== nothing
    n_inl, (KP, inl, err) = res
    fo["p3p_n"] = [n_inl]; fo["p3p_KP"] = Matrix{Float64}(KP); fo["p3p_inliers"] = UInt8.(inl); fo["p3p_error"] = [Float64(err)]
end
fp1 = [SLAM.Point2f(Pz["fp_px1"][i, :]...) for i in 1:size(Pz["fp_px1"], 1)]; fp2 = [SLAM.Point2f(Pz["fp_px2"][i, :]...) for i in 1:size(Pz["fp_px2"], 1)]
fd1 = [SLAM.Point2f(Pz["fp_pd1"][i, :]...) for i in 1:size(Pz["fp_pd1"], 1)]; fd2 = [SLAM.Point2f(Pz["fp_pd2"][i, :]...) for i in 1:size(Pz["fp_pd2"], 1)]
n5, (E5, P5, inl5, err5) = RecoverPose.five_point_ransac(fp1, fp2, fd1, fd2, K3(Pz["fp_K"]), K3(Pz["fp_K"]), cache; max_repr_error = 3.0)   # front_end.jl:305-308
fo["fp_n"] = [n5]; fo["fp_E"] = Matrix{Float64}(E5); fo["fp_P"] = Matrix{Float64}(P5); fo["fp_inliers"] = UInt8.(inl5)
npzwrite(joinpath(dir, "julia_frontend_v1.npz"), fo)
println("wrote ", joinpath(dir, "julia_frontend_v1.npz"), ": ", sum(present), " of ", n, " keypoints kept, ", sum(stereo2), " stereo matches, ", sum(tri3), " map points")
