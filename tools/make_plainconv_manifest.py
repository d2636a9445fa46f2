"""Pins everything about the default "nnUNet" model (PlainConvUNet) that the in-tree reference code determines, as a
golden manifest: tests/golden/plainconv_manifest.json.  Build container only (reads /root/reference).

The class body lives in the absent third-party package dynamic_network_architectures, so activations can
not be pinned; what CAN is pinned here from the reference's own code:
  * topology   - `get_pool_and_conv_props` imported as-is from
                 /root/reference/nnunetv2/experiment_planning/experiment_planners/network_topology.py:30-105, called the
                 way the planner calls it (default_experiment_planner.py:276-279: min edge = UNet_featuremap_min_edge_length,
                 max_numpool 999999);
  * arch kwargs - the dictionary literal of default_experiment_planner.py:284-305, re-evaluated here from the planner's
                 own constants (read from the class's __init__ by ast, the module itself needs packages that are absent);
  * derived    - parameter counts, output shapes and per-layer forward GFLOP by the formulas of SURVEY.md 8d, computed
                 here from the pinned topology only (no code of this repository is involved).
tests/test_plainconv_manifest.py asserts the oracle AND the product against the file.
"""
import ast
import importlib.util
import json
import os

import numpy as np

REF = "/root/reference/nnunetv2/experiment_planning/experiment_planners"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "plainconv_manifest.json")


def reference_topology_fn():
    spec = importlib.util.spec_from_file_location("ref_network_topology", os.path.join(REF, "network_topology.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.get_pool_and_conv_props


def planner_constants():
    """self.UNet_* literals assigned in ExperimentPlanner.__init__ (default_experiment_planner.py:28-80)"""
    tree = ast.parse(open(os.path.join(REF, "default_experiment_planner.py")).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Attribute) \
                and node.targets[0].attr.startswith("UNet_"):
            try:
                out[node.targets[0].attr] = ast.literal_eval(node.value)
            except ValueError:
                pass  # class references etc.
    return out


def build_case(name, spacing, patch, in_ch, classes, K, topo):
    dim = len(spacing)
    npool, strides, kernels, patch_out, divis = topo(spacing, patch, K["UNet_featuremap_min_edge_length"], 999999)
    n = len(strides)
    max_f = K["UNet_max_features_2d"] if dim == 2 else K["UNet_max_features_3d"]
    feats = [min(max_f, K["UNet_base_num_features"] * 2 ** i) for i in range(n)]
    nconv_e = list(K["UNet_blocks_per_stage_encoder"][:n])
    nconv_d = list(K["UNet_blocks_per_stage_decoder"][:n - 1])
    arch = {
        "n_stages": n, "features_per_stage": feats,
        "conv_op": f"torch.nn.modules.conv.Conv{dim}d",
        "kernel_sizes": [list(map(int, k)) for k in kernels], "strides": [list(map(int, s)) for s in strides],
        "n_conv_per_stage": nconv_e, "n_conv_per_stage_decoder": nconv_d, "conv_bias": True,
        "norm_op": f"torch.nn.modules.instancenorm.InstanceNorm{dim}d", "norm_op_kwargs": {"eps": 1e-5, "affine": True},
        "dropout_op": None, "dropout_op_kwargs": None, "nonlin": "torch.nn.LeakyReLU",
        "nonlin_kwargs": {"inplace": True},
    }
    # ---- derived figures (SURVEY.md 8d formulas) ----
    patch_out = [int(v) for v in patch_out]
    edges = list(patch_out)
    level_edges, params, gflop = [], 0, {}
    cin = in_ch
    for s in range(n):
        edges = [e // st for e, st in zip(edges, strides[s])]
        level_edges.append(list(edges))
        vox = int(np.prod(edges))
        k = int(np.prod(kernels[s]))
        for i in range(nconv_e[s]):
            params += cin * feats[s] * k + feats[s] + 2 * feats[s]
            gflop[f"enc{s}.{i}"] = 2.0 * vox * cin * feats[s] * k / 1e9
            cin = feats[s]
    for lvl in range(n - 2, -1, -1):
        below, skip = feats[lvl + 1], feats[lvl]
        up = int(np.prod(strides[lvl + 1]))
        params += below * skip * up + skip
        gflop[f"up{lvl}"] = 2.0 * int(np.prod(level_edges[lvl + 1])) * below * skip * up / 1e9
        k = int(np.prod(kernels[lvl]))
        c = 2 * skip
        for i in range(nconv_d[n - 2 - lvl]):
            params += c * skip * k + skip + 2 * skip
            gflop[f"dec{lvl}.{i}"] = 2.0 * int(np.prod(level_edges[lvl])) * c * skip * k / 1e9
            c = skip
        params += skip * classes + classes
        gflop[f"seg{lvl}"] = 2.0 * int(np.prod(level_edges[lvl])) * skip * classes / 1e9
    return {
        "name": name, "spacing": list(spacing), "initial_patch": list(map(int, patch)), "patch_size": patch_out,
        "input_channels": in_ch, "num_classes": classes,
        "num_pool_per_axis": [int(v) for v in npool], "shape_must_be_divisible_by": [int(v) for v in divis],
        "arch_kwargs": arch,
        "deep_supervision_output_shapes": [[classes] + level_edges[l] for l in range(n - 1)],  # highest resolution first
        "deep_supervision_scales": [[float(level_edges[l][a]) / patch_out[a] for a in range(dim)] for l in range(n - 1)],
        "parameter_count": int(params),
        "forward_gflop_per_sample": {k: round(v, 9) for k, v in gflop.items()},
        "forward_gflop_per_sample_total": round(sum(gflop.values()), 4),
    }


def main():
    topo = reference_topology_fn()
    K = planner_constants()
    need = ["UNet_base_num_features", "UNet_max_features_2d", "UNet_max_features_3d", "UNet_featuremap_min_edge_length",
            "UNet_blocks_per_stage_encoder", "UNet_blocks_per_stage_decoder"]
    assert all(k in K for k in need), K
    cases = [
        build_case("3d_fullres_128", (1.0, 1.0, 1.0), (128, 128, 128), 1, 2, K, topo),   # BASELINE configs[1] / [4]
        build_case("2d_512", (1.0, 1.0), (512, 512), 1, 2, K, topo),                      # BASELINE configs[0]
        build_case("3d_aniso_thick_slices", (3.0, 1.0, 1.0), (40, 192, 160), 2, 3, K, topo),
        build_case("3d_small_64", (1.0, 1.0, 1.0), (64, 64, 64), 1, 2, K, topo),
        build_case("2d_rect", (1.0, 1.0), (320, 256), 4, 4, K, topo),
    ]
    doc = {
        "generated_by": "tools/make_plainconv_manifest.py",
        "sources": {
            "topology": "nnunetv2/experiment_planning/experiment_planners/network_topology.py:30-105 (imported as-is)",
            "arch_kwargs": "nnunetv2/experiment_planning/experiment_planners/default_experiment_planner.py:284-305",
            "planner_constants": {k: K[k] for k in need},
        },
        "cases": cases,
    }
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
    for c in cases:
        print(c["name"], c["arch_kwargs"]["features_per_stage"], c["arch_kwargs"]["strides"], c["parameter_count"],
              c["forward_gflop_per_sample_total"])


if __name__ == "__main__":
    main()
