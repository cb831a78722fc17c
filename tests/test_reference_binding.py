"""The reference's own gaussian_renderer/__init__.py executed on top of the REAL drop-in module
(gaussian-garments_amd/diff_gaussian_rasterization_depth_alpha), with only ggsplat.rasterizer.rasterize_gaussians replaced by
a recorder: what arrives there must equal tests/golden/render_args.npz (tests/golden/check_reference_binding.py).
CPU suite, authoring container only: the reference tree never travels to the GPU box."""
import importlib.util
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isfile("/root/reference/gaussian_renderer/__init__.py"),
                                reason="the reference tree exists in the authoring container only")


def _checker():
    spec = importlib.util.spec_from_file_location("check_reference_binding", os.path.join(HERE, "golden", "check_reference_binding.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_reference_render_file_binds_the_product_module_and_hands_over_the_recorded_arguments():
    """render() in its five recorded scenarios and doll_render() in its four (gaussian_renderer/__init__.py:21-221): keyword-
    constructed settings (:39-52, :142-155), keyword call (:103-111, :208-216), the 'exactly one of' errors."""
    assert _checker().run() >= 40


def test_a_drifted_forward_signature_is_caught(monkeypatch):
    """The check has teeth: rename one keyword of the product's forward() and the reference's call (keywords, :103-111) fails."""
    m = _checker()
    import sys
    for p in (m.ROOT, m.PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import diff_gaussian_rasterization_depth_alpha as product

    def forward(self, means3D, means2D, opacities, sh=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        raise AssertionError("not reached")
    monkeypatch.setattr(product.GaussianRasterizer, "forward", forward)
    with pytest.raises(TypeError, match="shs"):
        m.run()


def test_sys_modules_are_left_as_found():
    import sys
    before = {k: sys.modules.get(k) for k in ("scene", "utils", "utils.sh_utils", "scene.gaussian_model")}
    _checker().load_reference_renderer()
    assert {k: sys.modules.get(k) for k in before} == before
