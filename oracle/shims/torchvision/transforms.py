"""Placeholder: dataset/*.py import torchvision.transforms; the hot path never calls it."""
