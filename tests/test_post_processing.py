"""Post-processing (reference src/post_processing.rs): the oracle's restatement on hand cases (CPU) and the
HIP kernel against the oracle, bit for bit (GPU)."""
import numpy as np
import pytest

from rustray_amd.flat import make_config
from tests.helpers import camera_for, load_scene


def _frame(h, w):
    rgba = np.zeros((h, w, 4), np.uint8); rgba[..., 3] = 255
    return rgba, np.zeros((h, w, 3), np.float32), np.zeros((h, w), np.uint32)


def test_outline_hand_case(oracle):
    rgba, nrm, ids = _frame(4, 5)
    rgba[..., :3] = 100
    ids[1:3, 1:3] = 7
    out = oracle.post_process(rgba, nrm, ids, cavity=False, outline=True)
    # interior pixel of a uniform 2x2 block has two equal neighbours of four: opacity 0.5 -> 127
    assert out[1, 1, :3].tolist() == [127, 127, 127]
    # a pixel whose four neighbours share its id keeps its colour (opacity 0 is not > 0)
    ids2 = np.zeros((5, 5), np.uint32)
    rg2, n2, _ = _frame(5, 5); rg2[..., :3] = 100
    out2 = oracle.post_process(rg2, n2, ids2, cavity=False, outline=True)
    assert out2[2, 2, :3].tolist() == [100, 100, 100]
    # top-left corner: the (0,-1) and (-1,0) neighbours fall outside the linear index range and read id 0 == own id
    assert out2[0, 0, :3].tolist() == [100, 100, 100]
    # linear-index quirk (:40-45): at the right border, x + 1 reads the first pixel of the NEXT row
    ids3 = np.zeros((3, 4), np.uint32); ids3[2, 0] = 9
    rg3, n3, _ = _frame(3, 4); rg3[..., :3] = 100
    out3 = oracle.post_process(rg3, n3, ids3, cavity=False, outline=True)
    assert out3[1, 3, :3].tolist() == [63, 63, 63]   # one of four neighbours differs: (2,0) seen through the wrap


def test_cavity_hand_case(oracle):
    rgba, nrm, ids = _frame(3, 3)
    rgba[..., :3] = 100
    nrm[2, 1, 2] = 0.2    # (0,+1) neighbour's z
    nrm[1, 2, 0] = 0.1    # (+1,0) neighbour's x
    out = oracle.post_process(rgba, nrm, ids, cavity=True, outline=False)
    d = np.float32(0.2) + np.float32(0.1)
    cur = np.float32(2.0) * (d * (np.float32(1.0) - d * np.float32(1.15)))   # ridge branch of curvature_soft_clamp
    assert out[1, 1, 0] == int(np.float32(100.0) * (cur + np.float32(1.0)))
    # NaN normals (all-miss pixels): comparisons are false, curvature = 2 * 0.25 / 1.15
    nrm[:] = np.nan
    out = oracle.post_process(rgba, nrm, ids, cavity=True, outline=False)
    k = np.float32(2.0) * (np.float32(0.25) / np.float32(1.15)) + np.float32(1.0)
    assert out[1, 1, 0] == int(np.float32(100.0) * k)
    rgba[..., :3] = 250
    out = oracle.post_process(rgba, nrm, ids, cavity=True, outline=False)
    assert out[1, 1, 0] == 255   # clamp


@pytest.mark.gpu
@pytest.mark.parametrize("cavity,outline", [(True, False), (False, True), (True, True), (False, False)])
def test_gpu_post_processing_bit_exact(hip, oracle, cavity, outline):
    fs = load_scene("spheres_room")
    cam = camera_for(fs, 161, 97).c_struct()   # odd sizes: border and wrap-around cases matter
    with hip.DeviceScene(fs, 0) as ds:
        fr = ds.render(cam, make_config(samples=2, monte_carlo=True, seed=5))
    got = hip.post_process(fr["rgba"], fr["normal"], fr["object_id"], cavity, outline)
    want = oracle.post_process(fr["rgba"], fr["normal"], fr["object_id"], cavity, outline)
    assert np.array_equal(got, want)
    if cavity or outline:
        assert not np.array_equal(got, fr["rgba"])


@pytest.mark.gpu
def test_gpu_post_processing_on_miss_pixels_and_device_path(hip, oracle):
    import torch
    fs = load_scene("spheres")   # plenty of all-miss pixels: NaN normals, id 0
    cam = camera_for(fs, 256, 256).c_struct()
    with hip.DeviceScene(fs, 0) as ds:
        fr = ds.render(cam, make_config(samples=1))
    assert np.isnan(fr["normal"]).any()
    want = oracle.post_process(fr["rgba"], fr["normal"], fr["object_id"], True, True)
    rg = torch.from_numpy(fr["rgba"]).cuda(); nr = torch.from_numpy(fr["normal"]).cuda()
    ids = torch.from_numpy(fr["object_id"].astype(np.int32)).cuda(); out = torch.empty_like(rg)
    hip.post_process_device(256, 256, True, True, rg.data_ptr(), nr.data_ptr(), ids.data_ptr(), out.data_ptr(), 0,
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    with pytest.raises(hip.RustrayHipError):
        hip.post_process_device(256, 256, True, True, rg.data_ptr(), 0, ids.data_ptr(), out.data_ptr(), 0)
