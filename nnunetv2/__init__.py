"""Import-only `nnunetv2` namespace over the MI355X-native hot path (package `nnuzoo_amd`).

The reference resolves its extension points by NAME at run time:
  * trainer plugins: `recursive_find_python_class(join(nnunetv2.__path__[0], "training", "nnUNetTrainer"), name,
    "nnunetv2.training.nnUNetTrainer")`  (/root/reference/nnunetv2/run/run_training.py:39-46,
    /root/reference/nnunetv2/inference/predict_from_raw_data.py:105-106);
  * networks: `pydoc.locate(plans[...]["network_class_name"])`
    (/root/reference/nnunetv2/utilities/get_network_from_plans.py:27) and `from nnunetv2.nets.<file> import ...` in the
    plugins.
This package provides those module paths - one module per reference file name, holding the same public names - and
every one of them re-exports the native implementation; no arithmetic lives here.  Modules of the reference that are
outside the hot path (planning, preprocessing, data loading, evaluation, image IO, CLI) are deliberately absent.
"""
__path__ = __import__("pkgutil").extend_path(__path__, __name__)  # a pip-installed distribution of the same name supplies every module this namespace does not define
