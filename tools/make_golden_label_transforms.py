"""Round-6 fixture for the label-side transforms of the training chain that the REFERENCE defines itself
(/root/reference/nnunetv2/training/data_augmentation/custom_transforms/: region_based_training.py ConvertSegmentationToRegionsTransform,
cascade_transforms.py MoveSegAsOneHotToData, masking.py MaskTransform), generated in the BUILD CONTAINER from those classes
(imported under tools/ref_shim.py; batchgenerators' AbstractTransform - absent - is `object`, the classes use nothing of it).
    python tools/make_golden_label_transforms.py
Stored: seeded int16 segmentations / float images (2-D and 3-D) and what the reference's transforms return for them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_shim  # noqa: E402


def main():
    ref_shim.install()
    import batchgenerators.transforms.abstract_transforms as at
    at.AbstractTransform = object
    import acvl_utils.morphology.morphology_helper  # noqa: F401  (mocked; cascade_transforms imports it at module level)
    from nnunetv2.training.data_augmentation.custom_transforms.cascade_transforms import MoveSegAsOneHotToData
    from nnunetv2.training.data_augmentation.custom_transforms.masking import MaskTransform
    from nnunetv2.training.data_augmentation.custom_transforms.region_based_training import ConvertSegmentationToRegionsTransform
    out = {}
    rng = np.random.RandomState(7)
    for tag, sp in (("2d", (37, 29)), ("3d", (9, 14, 11))):
        seg = rng.randint(-1, 5, size=(3, 2) + sp).astype(np.int16)            # labels -1 .. 4, two seg channels
        data = rng.randn(3, 2, *sp).astype(np.float32)
        regions = ((1, 2, 3), (2, 3), 3, (4,), -1)                               # nested regions, a bare int, the ignore label last
        out[f"seg_{tag}"], out[f"data_{tag}"] = seg, data
        out[f"regions_{tag}"] = ConvertSegmentationToRegionsTransform(regions, "seg", "seg", 0)(seg=seg.copy())["seg"]
        d = MoveSegAsOneHotToData(1, (1, 2, 4), "seg", "data", True)(data=data.copy(), seg=seg.copy())
        out[f"onehot_data_{tag}"], out[f"onehot_seg_{tag}"] = d["data"], d["seg"]
        out[f"masked_{tag}"] = MaskTransform([1], 0, 0, "data", "seg")(data=data.copy(), seg=seg.copy())["data"]
    path = os.path.join(ROOT, "tests", "golden", "label_transforms.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
