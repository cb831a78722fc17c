"""DIAGNOSTIC (variant library tools/dbg/variants/r06_alive_bbox_census.patch): of the forward's quadrant passes, how many would a test
"the splat's alpha AABB misses the bounding box of the quadrant's still-compositing pixels" skip -- all lists, and lists of >= 384."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-garments_amd"))
from ggsplat import _lib, rasterizer as R, synthetic as S
from ggsplat.mesh_gaussian_model import MeshGaussianModel
dev, W, H = "cuda", 1920, 1080
v, f = S.skirt_mesh(); m = MeshGaussianModel.from_tensors(v, f, S.skirt_gaussian_params(f.shape[0], 0), 0, device=dev)
with torch.no_grad():
    inp = (m.get_xyz, m.get_opacity, m.get_features, None, m.get_scaling, m.get_rotation, None)
L = _lib.lib()
for ci in (5, 40, 70, 100, 150):
    ck = S.stack_cameras(S.rig_cameras()[ci:ci + 1], device=dev)
    kw = dict(view=ck["view"], proj=ck["proj"], campos=ck["campos"], tanfov=ck["tanfov"], bg=torch.zeros(3, device=dev), W=W, H=H, sh_degree=0)
    R.forward_views(*inp, **kw, keep_state=False)
    sf = R.StagedForward(*inp, **kw)
    sf.run(sf.COUNT | sf.BIN)
    c = torch.zeros(6, dtype=torch.int64, device=dev)
    st = sf.state
    _lib.check(L.ggs_count_forward_visits(C.byref(st.prm), st.geom.data_ptr(), st.bin.data_ptr(), st.cap, c.data_ptr(),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "census")
    n = c.tolist()
    print(f"camera {ci}: quadrant passes {n[0]}, AABB misses the alive box in {n[3]} ({100 * n[3] / max(n[0], 1):.1f} %); lists >= 384: passes {n[4]}, "
          f"missed {n[5]} ({100 * n[5] / max(n[4], 1):.1f} %); passes that blend {n[1]}")
