"""Config plumbing with the reference's key names (reference utils/general_utils.py:62-69)."""
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "benchmarks")

# reference evaluate.py:118-124
BENCHMARK_CONFIGS = {
    "kitti_test": "test_kitti_config.yaml",
    "lokitti": "lokitti_config.yaml",
    "rotkitti": "rotkitti_config.yaml",
    "nuscenes_test": "test_nuscenes_config.yaml",
    "lonuscenes": "lonuscenes_config.yaml",
    "rotnuscenes": "rotnuscenes_config.yaml",
}


def update_namespace_from_yaml(args, yaml_path):
    """Flat YAML -> attributes on an argparse.Namespace-like object (same behaviour as the reference)."""
    with open(yaml_path, "r") as f:
        data = yaml.safe_load(f)
    for key, value in data.items():
        setattr(args, key, value)
    return args


def benchmark_config_path(benchmark):
    return os.path.join(CONFIG_DIR, BENCHMARK_CONFIGS[benchmark])
