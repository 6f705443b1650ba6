"""Shared helpers of the harnesses: project defaults, dataset front-end with device-side preprocessing, batched evaluation."""
