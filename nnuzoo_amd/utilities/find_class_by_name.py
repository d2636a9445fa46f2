"""Plugin discovery by class name - the reference's resolution rule
(/root/reference/nnunetv2/utilities/find_class_by_name.py:7-24): search the plain modules of `folder` first (importing
each as `<current_module>.<name>` until one has the attribute), then descend into sub-packages depth-first."""
import importlib
import os
import pkgutil


def recursive_find_python_class(folder: str, class_name: str, current_module: str):
    entries = list(pkgutil.iter_modules([folder]))
    for _, name, is_pkg in entries:
        if is_pkg:
            continue
        found = getattr(importlib.import_module(f"{current_module}.{name}"), class_name, None)
        if found is not None:
            return found
    for _, name, is_pkg in entries:
        if not is_pkg:
            continue
        found = recursive_find_python_class(os.path.join(folder, name), class_name, f"{current_module}.{name}")
        if found is not None:
            return found
    return None
