"""Checkpoint files in the reference's layout (src/dl/train.py:458-479 save_model, src/d_fine/utils.py:156-181
load_tuning_state): ``last.pt`` / ``model.pt`` are PLAIN ``state_dict``s of the EMA model when there is one (else of the
live model), with the reference's key names, shapes and dtypes -- a file written here loads ``strict=True`` into the
reference's ``build_model`` and the other way round (dfine.load_checkpoint_state also accepts its ``{"model": ...}`` and
``{"ema": {"module": ...}}`` wrappers). No optimizer / scheduler state: the reference has no resume either."""
from pathlib import Path

import torch


def export_state_dict(model):
    """``model.state_dict()`` with every tensor detached into its own storage (parameters here are views of flat
    buffers: saving the views would serialise the whole buffer behind each of them)."""
    return {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}


def save_model(train_step_or_model, path_to_save, metrics=None, best_metric=0.0, decision_metrics=("f1",)):
    """Mirror of Trainer.save_model: always writes ``last.pt``; writes ``model.pt`` when the mean of the decision metrics
    improved (or when no metrics are given). Returns the best metric so far."""
    model = train_step_or_model
    if hasattr(train_step_or_model, "model"):  # a TrainStep: prefer the EMA weights, like the reference
        ema = getattr(train_step_or_model, "ema", None)
        model = ema.model if ema is not None else train_step_or_model.model
    path = Path(path_to_save)
    path.mkdir(parents=True, exist_ok=True)
    sd = export_state_dict(model)
    torch.save(sd, path / "last.pt")
    if metrics is None:
        torch.save(sd, path / "model.pt")
        return best_metric
    decision = sum(float(metrics[m]) for m in decision_metrics) / len(decision_metrics)
    if decision > best_metric:
        torch.save(sd, path / "model.pt")
        return decision
    return best_metric
