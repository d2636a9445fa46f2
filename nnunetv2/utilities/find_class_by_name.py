"""`nnunetv2.utilities.find_class_by_name` of the reference (/root/reference/nnunetv2/utilities/find_class_by_name.py:7-24) -> native implementation in `nnuzoo_amd.utilities.find_class_by_name`."""
from nnuzoo_amd.utilities.find_class_by_name import recursive_find_python_class  # noqa: F401

__all__ = ['recursive_find_python_class']
