"""Device selection, same surface as aidial_rag/embeddings/detect_device.py:6-28,
with one member added for this build: ``rocm`` (the only device this package computes on)."""

from enum import Enum

from .. import _native as nat


class DeviceType(str, Enum):
    AUTO = "auto"
    CPU = "cpu"
    CUDA = "cuda"
    ROCM = "rocm"

    def __str__(self) -> str:
        return str(self.value)


def autodetect_device() -> DeviceType:
    return DeviceType.ROCM if nat.device_count() > 0 else DeviceType.CPU


def detect_device(device_str: str) -> DeviceType:
    if device_str == DeviceType.AUTO:
        return autodetect_device()
    try:
        return DeviceType(device_str)
    except ValueError:
        raise ValueError(f"Unknown device type: {device_str}")  # detect_device.py:26
