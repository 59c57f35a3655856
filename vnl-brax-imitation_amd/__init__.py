"""MI355X-native vectorised rodent-imitation rollout (hot path of talmolab/VNL-Brax-Imitation)."""
__version__ = "0.1.0"
