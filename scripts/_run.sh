for rep in 1 2; do
for L in libslamhip.so libslamhip_old.so; do
  SLAMHIP_LIB=$GRAFT_REPO_ROOT/slam.jl_amd/$L timeout 400 python bench.py --no-cpu --no-sweep --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$L', round(d['value']), round(d['pose']['frontend_with_pose']['value']), round(d['pose']['frontend_with_scene_pose_seams']['value']) if 'frontend_with_scene_pose_seams' in d['pose'] else '')"
done; done
