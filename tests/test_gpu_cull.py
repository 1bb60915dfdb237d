"""-m gpu: the conservative (group of 64 voxels, keyframe) culling in front of the observation pass (csrc/device/cull_kernels.hip) may only skip pairs whose
observation weight is zero for every voxel of the group: the Eg rows — keyframes, weights, residuals — must be BIT-IDENTICAL with and without it, on scenes that
exercise each of its tests (footprint outside the image, no valid depth under it, occlusion distance on / off, lens distortion, more than one mask word).
The same holds for the weight-bound prefilter inside k_observe (observe.hip), which skips a keyframe whose best possible weight cannot enter a voxel's top-n."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

DIST = np.array([0.08, -0.03, 0.002, 0.003, -0.002])

CASES = {
    # name: (scene keywords, distortion, optimizer-config keywords, least fraction of pairs that must be culled)
    "sphere": (dict(), None, dict(), 0.2),
    "zoomed_in": (dict(fx=320.0, cam_dist=0.16), None, dict(), 0.2),                   # most of the object outside every image
    "close_camera": (dict(radius_vox=12, cam_dist=0.075), None, dict(), 0.05),         # spheres near the camera plane: few decisions possible
    "distorted": (dict(), DIST, dict(), 0.2),
    "distorted_zoomed": (dict(fx=260.0, cam_dist=0.18), 3.0 * DIST, dict(), 0.1),
    "no_occlusion_test": (dict(fx=300.0, cam_dist=0.17), None, dict(occlusion_distance=0.0), 0.02),
    "tight_occlusion": (dict(), None, dict(occlusion_distance=0.003), 0.2),
    "two_mask_words": (dict(K=40, width=80, height=60), None, dict(num_observations=8), 0.2),
    "keep_all": (dict(K=4), None, dict(num_observations=4), 0.1),
    "coarse_level": (dict(levels=2), None, dict(rgbd_level=1), 0.2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_rows_bit_identical_with_and_without_culling(oracle, monkeypatch, name):
    kw, dist, ckw, least = CASES[name]
    sc = helpers.small_scene(seed=5, **kw)
    if dist is not None:
        sc["dist"] = np.asarray(dist, np.float64)
    g, fr, arrays, vsh, thres = helpers.oracle_setup(oracle, sc)
    cfg = helpers.gpu_cfg(helpers.oracle_cfg(oracle, thres, **ckw))
    ctx = helpers.gpu_context(sc, arrays, vsh)
    try:
        monkeypatch.setenv("I3D_NO_CULL", "1")
        ctx.debug_assemble(cfg, 0)
        pairs0, culled0 = ctx.debug_cull_stats()
        assert culled0 == -1
        ref = ctx.debug_eg_rows(jac=False)
        sizes0 = ctx.problem_sizes()
        for mode in (None, "2", "3"):          # both on (the default) / only the weight-bound prefilter of k_observe / only the group masks
            if mode is None:
                monkeypatch.delenv("I3D_NO_CULL")
            else:
                monkeypatch.setenv("I3D_NO_CULL", mode)
            ctx.debug_assemble(cfg, 0)
            pairs, culled = ctx.debug_cull_stats()
            got = ctx.debug_eg_rows(jac=False)
            assert ctx.problem_sizes() == sizes0 and sizes0["eg"] > 0
            for a, b, what in zip(got[:3], ref[:3], ("keyframe", "weight", "residual")):
                assert np.array_equal(a, b), (name, mode, what, int(np.sum(a != b)))
            if mode == "2":
                assert culled == -1
            else:
                assert pairs == pairs0 and culled >= least * pairs, (name, mode, pairs, culled)      # ... and the culling does cull
    finally:
        ctx.close()
