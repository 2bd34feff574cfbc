"""slam_jl_amd -- MI355X-native hot path of pxl-th/SLAM.jl behind SLAM.jl's own
function seams (extractor -> LK pyramid -> forward-backward LK -> local BA).

Host-side mirror of the reference interface over the C ABI of libslamhip.so
(include/slamhip.h).  Julia's `name!` functions are spelled `name_` here.
The Julia `ccall` shim with the same surface is julia/SLAMHip.jl.

There is no CPU fallback: every function here runs hand-written HIP kernels
and raises if the library / a HIP device is unavailable."""
from ._lib import Context, Event, SlamHipError, default_context, load, LIB_PATH  # noqa: F401
from .params import Camera, Params  # noqa: F401
from .extractor import Extractor, detect, detect_batch, describe, brief_pattern  # noqa: F401
from .optical_flow import (LKPyramid, LucasKanade, update_, copy_, deepcopy, has_gradients, fb_tracking_,  # noqa: F401
                           optical_flow_matching, optical_flow_matching_frame, PyramidBatch, optical_flow_matching_batch,
                           optical_flow_matching_batch_kept)
from .bundle_adjustment import LocalBACache, bundle_adjustment_, bundle_adjustment_batch_, BABatch, ba_plan_order, pnp_bundle_adjustment, pnp_bundle_adjustment_batch  # noqa: F401
from .triangulation import triangulate, projection_matrices  # noqa: F401
from .pose import p3p_ransac, five_point_ransac, draw_samples, p3p_ransac_batch, five_point_ransac_batch  # noqa: F401
from .kitti import KittyDataset  # noqa: F401
from .frontend import FrontEnd, FrontEndConfig  # noqa: F401
from .saver import ReplaySaver  # noqa: F401
from .keypoint_set import (KeypointSet, stream_params, pose_samples, pose_inputs, pose_samples5, pose_5pt_inputs,  # noqa: F401
                           pose_5pt_compose)
