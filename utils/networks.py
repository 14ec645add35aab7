"""Drop-in for the reference's ``utils.networks`` (/root/reference/utils/networks.py:4-12)."""
import torch


def continue_train(model, opt, model_path, load_weights):
    """Restore ``{'state_dict','optimizer','epoch'}`` checkpoints; returns (model, opt, next_epoch).
    Reference checkpoints also pickle the argparse Namespace under 'config' (train.py:116), hence
    weights_only=False."""
    start_epoch = 1
    if load_weights:
        state = torch.load(model_path, map_location='cpu', weights_only=False)
        if opt is not None and state.get('optimizer') is not None:
            opt.load_state_dict(state['optimizer'])
        model.load_state_dict(state['state_dict'])
        start_epoch = 1 + int(state['epoch'])
    return model, opt, start_epoch
