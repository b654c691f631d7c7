"""Stream ordering helper shared by the head (producer) and the input block (consumer); kept in its
own module so that features.py does not import prediction_task.py."""
import torch


def wait_pending_grad(param):
    """Orders the current stream after a gradient contribution still running on a side stream."""
    ev = getattr(param, "_t4r_pending", None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
        param._t4r_pending = None
