"""Mixture of ViT experts: dense ``MoEClassifier`` and the expert-parallel wrapper (one expert per rank, all-to-all of logits)."""
