"""Per-voice evaluation metrics on the device.

The reference logs, three times per epoch, what its GrooveEvaluator computes from ``model.predict`` (ref:evaluator.py:171-177,
522-525): ``get_hits_accuracies`` / ``get_velocity_errors`` / ``get_micro_timing_errors`` over the 9 voices of
``ROLAND_REDUCED_MAPPING`` -- after three ``.cpu()`` copies of the predictions.  Here the (N,32,27) prediction tensor never
leaves the GPU: one reduction (gt_voice_metrics, bitwise reproducible) yields 30 floats, and those are what goes to the host /
W&B.  The evaluator itself is un-vendored (GrooveEvaluator submodule), so the dictionary keys below are this package's own;
the quantities are: fraction of (sequence, step) cells whose hit equals the ground truth's, and mean squared velocity / offset
difference over all cells, per voice and averaged over voices.
"""
import torch

# voice order of hvo_sequence's ROLAND_REDUCED_MAPPING (9 voices; column j of each of the three 9-wide HVO blocks)
VOICES = ("KICK", "SNARE", "HH_CLOSED", "HH_OPEN", "TOM_3_LO", "TOM_2_MID", "TOM_1_HI", "CRASH", "RIDE")
GROUPS = (("Hits_Accuracy", 0), ("Velocity_MSE", 10), ("Offset_MSE", 20))


def voice_metrics(model, hvo_pred, hvo_gt):
    """-> dict of python floats: ``{Hits_Accuracy|Velocity_MSE|Offset_MSE}_{Overall|<voice>}`` (ONE 30-float D2H copy)."""
    out = model.engine.voice_metrics(hvo_pred, hvo_gt).tolist()
    res = {}
    for name, base in GROUPS:
        res[name + "_Overall"] = out[base]
        for j, v in enumerate(VOICES):
            res["%s_%s" % (name, v)] = out[base + 1 + j]
    return res


def evaluate(model, inputs, gt_hvo, use_thres=True, thres=0.5):
    """What the reference's ``log_eval`` does for the scalar metrics (ref:evaluator.py:516-525): predict the whole set on the
    device (chunked), reduce per voice on the device, return the dictionary."""
    pred = model.predict_hvo(inputs, use_thres=use_thres, thres=thres)
    return voice_metrics(model, pred, torch.as_tensor(gt_hvo))
