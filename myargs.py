"""Global flag namespace, parsed at import like the reference's myargs.py (flag names, defaults and
types follow /root/reference/myargs.py:10-136).  `from myargs import args` works unchanged; unknown
command-line words (pytest / torchrun options) are ignored instead of aborting the import."""
import argparse

_FLAGS = [
    # model
    ('model_name', 'Unet', str), ('arch_encoder', 'resnet18', str), ('num_classes', 4, int),
    ('class_probs', [0., 0., 0., 0.], list),
    # optimisation (training scripts only)
    ('optim', 'adam', str), ('lr', 0.0001, float), ('weight_decay', 0.0001, float), ('beta1', 0.9, float),
    ('beta2', 0.999, float), ('num_epoch', 2000, int), ('start_epoch', 1, int), ('batch_size', 30, int),
    ('workers', 10, int), ('gpu_ids', '0', str), ('loss', 'mse', str),
    # checkpoints
    ('eval_model_pth', 'data/models/model_resnet18_194.pt', str), ('train_model_pth', 'data/models/*.pt', str),
    ('model_save_pth', 'data/models', str), ('continue_train', False, bool), ('save_models', 1, int),
    ('validate_model', 1, int),
    # data locations
    ('raw_train_pth', 'data/bach/wsi', str), ('raw_val_pth', 'data/bach/wsi', str), ('wsi_mask_pth', 'data/test/wsi_mask', str),
    ('train_image_pth', 'data/train', str), ('val_image_pth', 'data/val', str), ('train_hr_image_pth', 'data/train_hr', str),
    ('val_hr_image_pth', 'data/val_hr', str), ('val_save_pth', 'data/val/out', str),
    # tiling
    ('tile_w', 512, int), ('tile_h', 512, int), ('tile_stride_w', 128, int), ('tile_stride_h', 128, int),
    ('scan_level', 2, int), ('scan_resize', 1, int),
    # dataset statistics
    ('dataset_mean', (0.485, 0.456, 0.406), list), ('dataset_std', (0.229, 0.224, 0.225), list),
    ('epsilon', 1e-8, float),
]

parser = argparse.ArgumentParser(allow_abbrev=False)     # (a host program's own --mode must not be read as an abbreviation of --model_name)
for _name, _default, _type in _FLAGS:
    parser.add_argument('--' + _name, default=_default, type=_type)

args, _ignored = parser.parse_known_args()
