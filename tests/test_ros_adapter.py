"""The ROS 1 adapter under ros/ against tests/ros_mock -- a mock of the handful of ROS types the adapter touches (NodeHandle +
XmlRpcValue parameters, sensor_msgs Image / CameraInfo, image_transport camera subscriber / publishers, tf listener,
resource_retriever, nodelet / pluginlib), NOT ROS.  It makes the sources meet a compiler and runs the camera callback the way
image_transport would: CPU tests = compile checks and the no-subscriber early exit; GPU tests = frames through
RosFilter::on_frame against the oracle (16UC1 and 32FC1, both outputs / mask only over the bit-packed path / padded rows).
Whether the adapter works against a real ROS installation is still unknown (ros/README.md)."""
import os
import subprocess

import numpy as np
import pytest

import golden_io
from bench_support import workloads as WL
from realtime_urdf_filter_amd.filter import depth_f32_to_u16, depth_u16_to_f32

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "ros_mock")
EXE = os.path.join(ROOT, "examples", "bin", "ros_adapter_harness")
INC = ["-I" + os.path.join(MOCK, "include"), "-I" + os.path.join(ROOT, "ros", "include"), "-I" + os.path.join(ROOT, "include")]


def harness():
    if not os.path.exists(EXE):
        subprocess.check_call([os.path.join(ROOT, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    return EXE


@pytest.mark.parametrize("source", ["ros_filter.cpp", "rtuf_node.cpp", "rtuf_nodelet.cpp"])
def test_adapter_sources_compile_against_the_mock(source):
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror"] + INC + [os.path.join(ROOT, "ros", "src", source)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def run(tmp_path, depth, encoding, mode, pad=0, env=None, size=(640, 480), intrinsics=("525", "525", "319.5", "239.5")):
    (tmp_path / "m.urdf").write_text(WL.EXAMPLE_URDF)
    depth.tofile(tmp_path / "d.bin")
    cmd = [harness(), str(tmp_path / "m.urdf"), str(tmp_path / "d.bin"), str(size[0]), str(size[1])] + list(intrinsics) + ["5.0", encoding,
           str(tmp_path / "o.depth"), str(tmp_path / "o.mask"), mode] + ([str(pad)] if pad else [])
    return subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})))


def test_nobody_listens_nothing_runs(tmp_path):
    """Parameters (table, ~models with an integer scale, ~camera_offset with integers among the numbers, robot_description found
    through searchParam) parse without a log line, the adapter subscribes, and with no subscriber on either output the callback
    returns before anything touches the GPU (this test runs without one)."""
    fx = golden_io.Fixture("example_urdf_640x480")
    r = run(tmp_path, np.nan_to_num(fx.depth, nan=0.0, posinf=0.0).astype(np.float32), "32FC1", "nobody")
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "published depth 0 mask 0" in r.stdout and "log " not in r.stdout


def test_missing_required_parameter_is_reported(tmp_path):
    """A required private parameter that is absent is reported (ROS_FATAL in the reference too, src/urdf_filter.cpp:63-75)."""
    fx = golden_io.Fixture("example_urdf_640x480")
    r = run(tmp_path, fx.depth.astype(np.float32), "32FC1", "nobody", env={"RTUF_MOCK_DROP_PARAM": "~camera_frame"})
    assert "log FATAL: private parameter ~camera_frame is required" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.gpu
def test_callback_16uc1_both_outputs(tmp_path):
    fx = golden_io.Fixture("example_urdf_640x480")
    mm = depth_f32_to_u16(np.nan_to_num(fx.depth, nan=0.0, posinf=0.0))
    r = run(tmp_path, mm, "16UC1", "both")
    assert r.returncode == 0 and "published depth 1 mask 1" in r.stdout and "log " not in r.stdout, (r.stdout, r.stderr)
    masked16 = np.fromfile(tmp_path / "o.depth", np.uint16).reshape(480, 640)
    mask = np.fromfile(tmp_path / "o.mask", np.uint8).reshape(480, 640)
    import oracle.bindings as O
    om, ok = O.filter_frame(depth_u16_to_f32(mm), fx.projection, fx.draws, fx.offset_inv, fx.cam_tf, max_diff=0.05, replace_value=5.0)
    assert np.array_equal(mask, ok) and np.array_equal(masked16, depth_f32_to_u16(om)) and mask.any()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,pad", [("mask_only", 12), ("depth_only", 0), ("both", 40)])
def test_callback_32fc1_modes_and_padded_rows(tmp_path, mode, pad):
    """32FC1; only output_mask has a subscriber (one bit per pixel crosses the bus), only output_depth has one, both with an
    input image whose rows are padded (step > width * 4)."""
    fx = golden_io.Fixture("example_urdf_640x480")
    r = run(tmp_path, fx.depth.astype(np.float32), "32FC1", mode, pad)
    assert r.returncode == 0 and "log " not in r.stdout, (r.stdout, r.stderr)
    masked = np.fromfile(tmp_path / "o.depth", np.float32).reshape(480, 640) if mode != "mask_only" else None
    mask = np.fromfile(tmp_path / "o.mask", np.uint8).reshape(480, 640) if mode != "depth_only" else None
    if masked is not None and mask is not None:
        fx.check(masked, mask)
    elif mask is not None:
        assert np.array_equal(mask, fx.mask)
    else:
        assert np.array_equal(masked.view(np.uint32), fx.expected_masked().view(np.uint32))


@pytest.mark.parametrize("env,what", [({"RTUF_MOCK_SHORT_IMAGE": "1000"}, "malformed image"), ({"RTUF_MOCK_INFO_WIDTH": "320"}, "camera_info is for 320x480")])
def test_malformed_messages_are_refused_before_anything_is_read(tmp_path, env, what):
    """A sensor_msgs/Image whose data vector is shorter than step * height, or whose camera_info belongs to another image
    size, is reported and dropped (cv_bridge did that for the reference): no row of it is read, nothing is published, and the
    GPU is never touched (this test runs without one)."""
    fx = golden_io.Fixture("example_urdf_640x480")
    r = run(tmp_path, fx.depth.astype(np.float32), "32FC1", "both", env=env)
    assert "log ERROR: input_depth: " + what in r.stdout and "published depth 0 mask 0" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.gpu
def test_camera_info_without_a_size_is_filtered_like_the_reference_would(tmp_path):
    """Some drivers leave camera_info.width / height at 0; the reference reads only K / P from the message
    (src/urdf_filter.cpp:459-501) and filters such frames: so does the adapter, with the image's own size."""
    fx = golden_io.Fixture("example_urdf_640x480")
    r = run(tmp_path, fx.depth.astype(np.float32), "32FC1", "both", env={"RTUF_MOCK_INFO_UNSIZED": "1"})
    assert r.returncode == 0 and "published depth 1 mask 1" in r.stdout and "log " not in r.stdout, (r.stdout, r.stderr)
    fx.check(np.fromfile(tmp_path / "o.depth", np.float32).reshape(480, 640), np.fromfile(tmp_path / "o.mask", np.uint8).reshape(480, 640))


@pytest.mark.gpu
@pytest.mark.parametrize("encoding,mode", [("16UC1", "both"), ("16UC1", "mask_only"), ("32FC1", "mask_only"), ("32FC1", "both")])
def test_callback_width_that_is_not_a_multiple_of_four(tmp_path, encoding, mode):
    """A 322-pixel-wide camera: the fused 16UC1 kernels and the bit-packed mask need width % 4 == 0, so the facade takes such
    frames through the full 32FC1 planes and converts on the host with the reference's arithmetic -- every frame is
    published, bit-identical to the oracle (these frames used to throw and never publish)."""
    import oracle.bindings as O
    import scenes as S
    W, H = 322, 242
    fx = golden_io.Fixture("example_urdf_640x480")
    depth = S.sensor_depth(W, H, 3.0)
    f, cx, cy = 525.0 * W / 640, (W - 1) / 2, (H - 1) / 2
    P = S.projection(f, f, cx, cy, W, H)
    data = depth_f32_to_u16(np.nan_to_num(depth, nan=0.0, posinf=0.0)) if encoding == "16UC1" else depth.astype(np.float32)
    r = run(tmp_path, data, encoding, mode, size=(W, H), intrinsics=(repr(f), repr(f), repr(cx), repr(cy)))
    assert r.returncode == 0 and "log " not in r.stdout, (r.stdout, r.stderr)
    d32 = depth_u16_to_f32(data) if encoding == "16UC1" else data
    om, ok = O.filter_frame(d32, P, fx.draws, fx.offset_inv, fx.cam_tf, max_diff=0.05, replace_value=5.0)
    mask = np.fromfile(tmp_path / "o.mask", np.uint8).reshape(H, W)
    assert np.array_equal(mask, ok) and mask.any()
    if mode == "both":
        if encoding == "16UC1":
            assert np.array_equal(np.fromfile(tmp_path / "o.depth", np.uint16).reshape(H, W), depth_f32_to_u16(om))
        else:
            assert np.array_equal(np.fromfile(tmp_path / "o.depth", np.uint32).reshape(H, W), om.view(np.uint32))
