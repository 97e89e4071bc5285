from gemnet_pytorch_amd.training.metrics import MeanMetric, Metrics, BestMetrics  # noqa: F401
