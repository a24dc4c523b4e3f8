"""Helpers for the libuhdr.so facade tests: run the reference's own sample app, linked unmodified
against libultrahdr_amd/lib/libuhdr.so, as a subprocess."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libultrahdr_amd", "lib")
APP = os.path.join(LIBDIR, "ultrahdr_app")
FACADE = os.path.join(LIBDIR, "libuhdr.so")

# the 43 extern "C" functions of ultrahdr_api.h:301-905 (42 uhdr_* + is_uhdr_image), SURVEY.md 8b
API_SYMBOLS = """is_uhdr_image uhdr_add_effect_crop uhdr_add_effect_mirror uhdr_add_effect_resize uhdr_add_effect_rotate
uhdr_create_decoder uhdr_create_encoder uhdr_dec_get_base_image uhdr_dec_get_exif uhdr_dec_get_gainmap_height
uhdr_dec_get_gainmap_image uhdr_dec_get_gainmap_metadata uhdr_dec_get_gainmap_width uhdr_dec_get_icc
uhdr_dec_get_image_height uhdr_dec_get_image_width uhdr_dec_probe uhdr_dec_set_image uhdr_dec_set_out_color_transfer
uhdr_dec_set_out_img_format uhdr_dec_set_out_max_display_boost uhdr_decode uhdr_enable_gpu_acceleration
uhdr_enc_set_compressed_image uhdr_enc_set_exif_data uhdr_enc_set_gainmap_gamma uhdr_enc_set_gainmap_image
uhdr_enc_set_gainmap_scale_factor uhdr_enc_set_min_max_content_boost uhdr_enc_set_output_format uhdr_enc_set_preset
uhdr_enc_set_quality uhdr_enc_set_raw_image uhdr_enc_set_target_display_peak_brightness
uhdr_enc_set_using_multi_channel_gainmap uhdr_encode uhdr_get_decoded_gainmap_image uhdr_get_decoded_image
uhdr_get_encoded_stream uhdr_release_decoder uhdr_release_encoder uhdr_reset_decoder uhdr_reset_encoder""".split()


def built():
    return os.path.isfile(APP) and os.path.isfile(FACADE)


class Stages(dict):
    """The facade's stage table of one process (uhdr_hip_seam_stats, include/uhdr_hip.h): {stage: {"device": n, "reference": n, ...}}."""

    def on(self, where="device"):
        return sorted(k for k, v in self.items() if v[where] > 0 and k not in ("uhdr_call", "gainmap_copy_deferred", "gainmap_image_asked_for"))

    def n(self, stage, where="device"):
        return self.get(stage, {}).get(where, 0)


def run_app(args, gpu, cwd, timeout=600, env_extra=None):
    """-> (returncode, stdout, stderr, Stages: where every stage of the run's accelerated calls ran).  The app is the reference's own
    binary: the library writes its stage table to UHDR_HIP_SEAM_STATS_FILE when the process exits -- nothing is parsed off stderr."""
    import json
    import tempfile

    env = dict(os.environ)
    env.pop("UHDR_HIP_SEAM_TRACE", None)
    fd, stats_path = tempfile.mkstemp(suffix=".json", dir=cwd)
    os.close(fd)
    env["UHDR_HIP_SEAM_STATS_FILE"] = stats_path
    env.update(env_extra or {})
    cmd = [APP] + [str(a) for a in args] + (["-u", "1"] if gpu else [])
    p = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
    try:
        with open(stats_path) as f:
            text = f.read()
        stages = Stages(json.loads(text) if text.strip() else {})
    finally:
        os.unlink(stats_path)
    return p.returncode, p.stdout, p.stderr, stages


def encode_api1(p010_path, yuv_path, w, h, out, gpu, cwd, extra=(), env_extra=None):
    """BASELINE config 1's command line: P010 (P3... here BT.2100 / HLG / narrow) + YUV420 (BT.709) -> UltraHDR JPEG."""
    return run_app(["-m", 0, "-p", p010_path, "-y", yuv_path, "-w", w, "-h", h, "-a", 0, "-b", 1, "-C", 2, "-c", 0, "-t", 1,
                    "-R", 0, "-z", out] + list(extra), gpu, cwd, env_extra=env_extra)


def decode(jpg, ct, fmt, out, gpu, cwd, env_extra=None):
    return run_app(["-m", 1, "-j", jpg, "-o", ct, "-O", fmt, "-z", out], gpu, cwd, env_extra=env_extra)


def read(path):
    return np.fromfile(path, dtype=np.uint8)
