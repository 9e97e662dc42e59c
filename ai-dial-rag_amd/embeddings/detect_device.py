"""Which device the encoder and the indexes run on.

Keeps the reference's names (``DeviceType``, ``autodetect_device``, ``detect_device``;
aidial_rag/embeddings/detect_device.py:6-28) and its error text, and adds the one member this
build computes on: ``rocm``.  ``cpu`` and ``cuda`` stay valid *values* - configuration written for
the reference must still parse - but nothing in this package runs on them: the retrievers and the
encoder raise when no gfx950 device is present.
"""

from enum import Enum

from .. import _native as nat

_KNOWN = ("auto", "cpu", "cuda", "rocm")


class DeviceType(str, Enum):
    AUTO, CPU, CUDA, ROCM = _KNOWN

    def __str__(self) -> str:
        return self.value

    @property
    def computes_here(self) -> bool:
        """True for the device kind libmiretr.so has kernels for."""
        return self is DeviceType.ROCM


def autodetect_device() -> DeviceType:
    """``rocm`` when libmiretr sees at least one GPU, else ``cpu`` (on which this package refuses to compute)."""
    have_gpu = nat.device_count() > 0
    return DeviceType.ROCM if have_gpu else DeviceType.CPU


def detect_device(device_str: str) -> DeviceType:
    """``"auto"`` -> autodetect; otherwise the named member.  Unknown names raise the reference's ValueError."""
    name = str(device_str)
    if name not in _KNOWN:
        raise ValueError(f"Unknown device type: {device_str}")  # same message as detect_device.py:26
    return autodetect_device() if name == "auto" else DeviceType(name)
