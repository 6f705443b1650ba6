"""ViT fine-tuning entry points: ``model.ViT`` (factory), ``finetune`` (hand-rolled loop), ``ntrain`` (Lightning-shaped harness)."""
