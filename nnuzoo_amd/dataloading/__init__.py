from .device_loader import DeviceCaseStore, nnUNetDataLoader  # noqa: F401
