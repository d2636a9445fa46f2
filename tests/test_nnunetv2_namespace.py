"""The reference resolves trainers and networks by NAME.  These tests apply the reference's own
resolution rules - `recursive_find_python_class` over `<nnunetv2>/training/nnUNetTrainer` as run_training.py:39-46 and
predict_from_raw_data.py:105-106 call it, `pydoc.locate` on the class string the planner writes into plans.json
(get_network_from_plans.py:27), and the `from nnunetv2.<...> import <name>` lines of the plugins - to this repository and
check that each one lands on the MI355X-native implementation.  CPU only: nothing is launched."""
import importlib
import os
import pkgutil
import pydoc

import pytest
import torch

import nnunetv2
from nnuzoo_amd.synthetic import nnunet_plans


def reference_rule_find(folder: str, class_name: str, current_module: str):
    """independent restatement of /root/reference/nnunetv2/utilities/find_class_by_name.py:7-24 (modules of the folder
    first, then sub-packages) - deliberately not the product's own function, which is checked against it below"""
    hit = None
    for _, modname, ispkg in pkgutil.iter_modules([folder]):
        if not ispkg:
            m = importlib.import_module(current_module + "." + modname)
            if hasattr(m, class_name):
                hit = getattr(m, class_name)
                break
    if hit is None:
        for _, modname, ispkg in pkgutil.iter_modules([folder]):
            if ispkg:
                hit = reference_rule_find(os.path.join(folder, modname), class_name, current_module + "." + modname)
            if hit is not None:
                break
    return hit


TRAINER_FOLDER = os.path.join(nnunetv2.__path__[0], "training", "nnUNetTrainer")
# trainer name -> (module file name in the reference's plugin folder, native class)
TRAINERS = {
    "nnUNetTrainer": ("nnUNetTrainer", "nnuzoo_amd.training.nnUNetTrainer.nnUNetTrainer"),
    "nnUNetTrainerM2Net": ("nnUNetTrainerM2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerM2Net"),
    "nnUNetTrainerM2NetP": ("nnUNetTrainerM2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerM2NetP"),
    "nnUNetTrainerSwT2Net": ("nnUNetTrainerSwT2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerSwT2Net"),
    "nnUNetTrainerSSND2Net": ("nnUNetTrainerSSND2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerSSND2Net"),
    "nnUNetTrainerSSND2NetP": ("nnUNetTrainerSSND2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerSSND2NetP"),
    "nnUNetTrainerMambaND2Net": ("nnUNetTrainerMambaND2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerMambaND2Net"),
    "nnUNetTrainerMambaND2NetP": ("nnUNetTrainerMambaND2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerMambaND2NetP"),
    "nnUNetTrainerUNETR2Net": ("nnUNetTrainerUNETR2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerUNETR2Net"),
    "nnUNetTrainerLightMamba2Net": ("nnUNetTrainerLightMamba2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerLightMamba2Net"),
    "nnUNetTrainerLightMamba2NetP": ("nnUNetTrainerLightMamba2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerLightMamba2NetP"),
    "nnUNetTrainerLightMUNet": ("nnUNetTrainerLightMUNet", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerLightMUNet"),
    "nnUNetTrainerLM2Net": ("nnUNetTrainerLM2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerLM2Net"),
    "nnUNetTrainerLM2NetP": ("nnUNetTrainerLM2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerLM2NetP"),
    "nnUNetTrainerU2Net": ("nnUNetTrainerU2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerU2Net"),
    "nnUNetTrainerU2NetP": ("nnUNetTrainerU2Net", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerU2NetP"),
    "nnUNetTrainerU2NetMulti": ("nnUNetTrainerU2NetMulti", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerU2NetMulti"),
    "nnUNetTrainerU2NetMultiP": ("nnUNetTrainerU2NetMulti", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerU2NetMultiP"),
    "nnUNetTrainerSwUNETR": ("nnUNetTrainerSwUNETR", "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerSwUNETR"),
    "nnUNetTrainerSwinTransformerUnet": ("nnUNetTrainerSwinTransformerUnet",
                                         "nnuzoo_amd.training.zoo_trainers.nnUNetTrainerSwinTransformerUnet"),
}
TRAINERS = {k: v for k, v in TRAINERS.items()
            if os.path.exists(os.path.join(TRAINER_FOLDER, v[0] + ".py"))}  # families not built yet have no module


@pytest.mark.parametrize("name", sorted(TRAINERS))
def test_trainer_plugin_discovery(name):
    from nnunetv2.training.nnUNetTrainer.nnUNetTrainer import nnUNetTrainer as base
    from nnunetv2.utilities.find_class_by_name import recursive_find_python_class
    module_file, native = TRAINERS[name]
    cls = reference_rule_find(TRAINER_FOLDER, name, "nnunetv2.training.nnUNetTrainer")
    assert cls is not None and cls is pydoc.locate(native)
    assert cls is recursive_find_python_class(TRAINER_FOLDER, name, "nnunetv2.training.nnUNetTrainer")
    assert issubclass(cls, base)                                      # run_training.py:46
    assert cls.__name__ == name                                       # stored as checkpoint['trainer_name'] (:1309)
    assert hasattr(importlib.import_module(f"nnunetv2.training.nnUNetTrainer.{module_file}"), name)
    # get_trainer_from_args (run_training.py:65-66) constructs by keyword
    dim = 3 if name == "nnUNetTrainer" else 2
    plans, cfg, dj = nnunet_plans(dim, (32,) * dim, batch_size=2)
    tr = cls(plans=plans, configuration=cfg, fold=0, dataset_json=dj, unpack_dataset=True, device=torch.device("cpu"))
    assert tr.my_init_kwargs["configuration"] == cfg and tr.my_init_kwargs["fold"] == 0
    assert callable(tr.build_network_architecture) and callable(tr._get_deep_supervision_scales)


def test_unknown_trainer_is_none():
    assert reference_rule_find(TRAINER_FOLDER, "nnUNetTrainerDoesNotExist", "nnunetv2.training.nnUNetTrainer") is None


def test_plans_network_class_string_resolves_to_the_native_class():
    plans, cfg, _ = nnunet_plans(3, (128, 128, 128))
    name = plans["configurations"][cfg]["architecture"]["network_class_name"]
    assert name == "dynamic_network_architectures.architectures.unet.PlainConvUNet"   # what the planner writes
    from nnuzoo_amd.nets.plain_conv_unet import PlainConvUNet
    assert pydoc.locate(name) is PlainConvUNet
    # the reference's factory with the reference's argument list, CPU construction only
    from nnunetv2.utilities.get_network_from_plans import get_network_from_plans
    arch = plans["configurations"][cfg]["architecture"]
    net = get_network_from_plans(name, arch["arch_kwargs"], arch["_kw_requires_import"], 1, 2, allow_init=True,
                                 deep_supervision=True)
    assert isinstance(net, PlainConvUNet) and net.decoder.deep_supervision is True
    assert sum(p.numel() for p in net.parameters()) == 31_195_594


IMPORT_LINES = [
    # (module, names) exactly as the reference's files import them
    ("nnunetv2.nets.m2net", ["get_m2net_from_plans", "get_m2netp_from_plans", "M2Net", "M2NetP", "SS2D", "RSU4F",
                             "REBNCONV"]),                                     # nnUNetTrainerM2Net.py:9
    ("nnunetv2.nets.swt2net", ["get_swt2net_from_plans", "SwT2Net", "WindowAttention", "SwinTransformerBlock",
                               "SwinTransformerUnet", "Mlp"]),                # nnUNetTrainerSwT2Net.py:9
    ("nnunetv2.nets.ssnd2net", ["get_ssnd2net_from_plans", "SSND2Net", "SSND2NetP", "SSND", "GSC"]),
    ("nnunetv2.nets.u2net", ["get_u2net_from_plans", "get_u2netp_from_plans", "U2NET", "U2NETP", "RSU7", "RSU4F"]),   # nnUNetTrainerU2Net.py:9-10
    ("nnunetv2.nets.swt", ["get_swin_transformer_unet", "SwinTransformerUnet", "WindowAttention"]),
    ("nnunetv2.nets.seg_mamba.mamba_simple", ["Mamba"]),
    ("nnunetv2.nets.seg_mamba.selective_scan_interface", ["selective_scan_fn", "mamba_inner_fn"]),
    ("nnunetv2.training.loss.compound_losses", ["DC_and_CE_loss", "DC_and_BCE_loss"]),     # nnUNetTrainer.py:52
    ("nnunetv2.training.loss.deep_supervision", ["DeepSupervisionWrapper"]),               # :53
    ("nnunetv2.training.loss.dice", ["MemoryEfficientSoftDiceLoss"]),                      # :54
    ("nnunetv2.training.loss.robust_ce_loss", ["RobustCrossEntropyLoss"]),
    ("nnunetv2.training.lr_scheduler.polylr", ["PolyLRScheduler"]),                        # :55
    ("nnunetv2.utilities.get_network_from_plans", ["get_network_from_plans"]),             # :62
    ("nnunetv2.utilities.network_initialization", ["InitWeights_He"]),
    ("nnunetv2.utilities.ddp_allgather", ["AllGatherGrad"]),
    ("nnunetv2.utilities.helpers", ["softmax_helper_dim1"]),
    ("nnunetv2.utilities.find_class_by_name", ["recursive_find_python_class"]),
    ("nnunetv2.inference.predict_from_raw_data", ["nnUNetPredictor"]),
    ("nnunetv2.inference.sliding_window_prediction", ["compute_gaussian", "compute_steps_for_sliding_window"]),
]


@pytest.mark.parametrize("module,names", IMPORT_LINES, ids=[m for m, _ in IMPORT_LINES])
def test_reference_import_lines(module, names):
    m = importlib.import_module(module)
    for n in names:
        obj = getattr(m, n)
        assert obj.__module__.startswith("nnuzoo_amd."), (module, n, obj.__module__)


def test_namespace_modules_are_import_only():
    """no arithmetic in the namespace package: every module is a docstring + imports """
    import ast
    root = nnunetv2.__path__[0]
    for dirpath, _, files in os.walk(root):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in tree.body:
                ok = isinstance(node, (ast.Import, ast.ImportFrom)) or \
                    (isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant)) or \
                    (isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("__all__", "__path__"))
                assert ok, (f, ast.dump(node)[:80])


def test_namespace_defers_to_an_installed_distribution_for_other_modules(tmp_path):
    """modules this namespace does not define resolve from a distribution of the same name further down sys.path
    (pkgutil.extend_path) instead of failing with ModuleNotFoundError; the native classes keep precedence"""
    import subprocess
    import sys
    site = tmp_path / "site"
    for pkg, sub, mod, body in [("nnunetv2", "", "paths", "nnUNet_raw = 'from-the-installed-package'\n"),
                                ("nnunetv2", "nets", "only_upstream", "X = 7\n"),
                                ("dynamic_network_architectures", "building_blocks", "helper", "Y = 9\n")]:
        d = site / pkg / sub
        d.mkdir(parents=True, exist_ok=True)
        for q in (site / pkg, d):
            (q / "__init__.py").touch()
        (d / f"{mod}.py").write_text(body)
    code = ("import sys; sys.path.insert(0, %r); sys.path.append(%r)\n"
            "import nnunetv2.paths, nnunetv2.nets.only_upstream, dynamic_network_architectures.building_blocks.helper as h\n"
            "from nnunetv2.nets.m2net import M2Net\n"
            "import pydoc\n"
            "cls = pydoc.locate('dynamic_network_architectures.architectures.unet.PlainConvUNet')\n"
            "print(nnunetv2.paths.nnUNet_raw, nnunetv2.nets.only_upstream.X, h.Y, M2Net.__module__, cls.__module__)\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(site)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["from-the-installed-package", "7", "9", "nnuzoo_amd.nets.m2net",
                                  "nnuzoo_amd.nets.plain_conv_unet"], out.stdout
