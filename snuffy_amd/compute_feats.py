"""MI355X-native mirror of the feature-extraction stage (reference compute_feats.py): tile directory -> ViT embedder ->
one CSV of %.4f features per slide.

Kept (same names / meaning): BagDataset, bag_dataset, compute_feats, get_embedder_backbone, get_embedder,
_load_model_weights (positional key mapping of the DINO 'teacher' / MAE 'model' checkpoint, compute_feats.py:449-490),
the CSV format of compute_feats.py:256-266.  The embedder is snuffy_amd.vit (HIP kernels); tile decoding / resizing is
host work (PIL) exactly as in the reference -- SURVEY.md 8f ranks it "next" (batched / on-device preprocessing).
"""
import glob
import os
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import vit

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
VIT_BACKBONES = ("vit_tiny", "vit_small", "vit_base")


class BagDataset:
    """Tiles of one slide (reference compute_feats.py:66-101): returns {'input', 'label', 'position'}."""

    def __init__(self, files_list, transform=None, patch_labels_dict=None):
        self.files_list = files_list
        self.transform = transform
        self.patch_labels = patch_labels_dict or {}

    def __len__(self):
        return len(self.files_list)

    def __getitem__(self, idx):
        from PIL import Image
        path = self.files_list[idx]
        img = Image.open(path)
        patch_address = os.path.join(*path.split(os.path.sep)[-3:])           # class_name/bag_name/patch_name.jpeg
        label = self.patch_labels.get(patch_address, -1)
        row, col = Path(path).stem.split('-')[0].split('_')                    # "{row}_{col}[-17].jpeg"
        sample = {'input': img, 'label': label, 'position': np.asarray([int(row), int(col)])}
        return self.transform(sample) if self.transform else sample


class TileTransform:
    """Resize(224) (PIL bilinear with antialias, as torchvision's VF.resize on a PIL image), ToTensor, optional ImageNet
    normalisation (reference compute_feats.py:104-152, 173-177)."""

    def __init__(self, resize=224, normalize=False):
        self.resize, self.normalize = resize, normalize

    def __call__(self, sample):
        from PIL import Image
        img = sample['input'].convert('RGB')
        if self.resize is not None:
            w, h = img.size
            if (w <= h and w != self.resize) or (h < w and h != self.resize):   # shorter side -> resize (VF.resize(int))
                if w <= h:
                    img = img.resize((self.resize, int(self.resize * h / w)), Image.BILINEAR)
                else:
                    img = img.resize((int(self.resize * w / h), self.resize), Image.BILINEAR)
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        if self.normalize:
            t = (t - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(IMAGENET_STD).view(3, 1, 1)
        label = sample['label']
        assert isinstance(label, int), f"A sample label should be of type int, but {type(label)} received."
        return {**sample, 'input': t, 'label': torch.tensor(label)}


class DecodeOnly:
    """Host half of the batched preprocessing: decode the tile to RGB uint8 [H, W, 3] and stop -- Resize(224),
    ToTensor and the normalisation run on the GPU for the whole batch (snuffy_amd.tiles.preprocess_tiles, bit-identical to
    TileTransform).  The uint8 tile is a quarter of the bytes of the fp32 tensor on the PCIe link as well."""

    def __call__(self, sample):
        img = sample['input'].convert('RGB')
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
        label = sample['label']
        assert isinstance(label, int), f"A sample label should be of type int, but {type(label)} received."
        return {**sample, 'input': t, 'label': torch.tensor(label)}


def device_preprocess(args):
    """Batched on-device tile preprocessing is the default for the ViT backbones; --device_preprocess 0 keeps the reference's
    per-tile PIL path on the DataLoader workers."""
    is_vit = args.backbone in VIT_BACKBONES or args.backbone == 'vitbasetimm'
    return is_vit and int(getattr(args, 'device_preprocess', 1)) == 1


def bag_dataset(args, patches, patch_labels_dict=None):
    """(DataLoader, number of tiles) for one slide directory (reference compute_feats.py:155-197)."""
    is_vit = args.backbone in VIT_BACKBONES or args.backbone == 'vitbasetimm'
    if device_preprocess(args):
        tf = DecodeOnly()
    else:
        tf = TileTransform(224 if is_vit else None, normalize=(getattr(args, 'transform', 0) == 1))
    ds = BagDataset(files_list=patches, transform=tf, patch_labels_dict=patch_labels_dict)
    return DataLoader(ds, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers, drop_last=False), len(ds)


def get_embedder_backbone(args):
    """DINO / DINO-adapter / MAE-adapter ViT (reference compute_feats.py:369-438); parameters frozen."""
    kind = args.embedder
    if 'MAE' in kind:
        model = vit.mae_adapter_encoder(patch_size=16, embed_dim=768, depth=12, num_heads=12,
                                        adapter_ffn_scalar=str(getattr(args, 'adapter_ffn_scalar', '1.0')),
                                        adapter_ffn_num=getattr(args, 'ffn_num', 64), adapter_d_model=768)
        num_feats = 768
    else:
        if args.backbone not in VIT_BACKBONES:
            raise ValueError(f"Unknow architecture: {args.backbone}")
        ctor = getattr(vit, args.backbone)
        width = {'vit_tiny': 192, 'vit_small': 384, 'vit_base': 768}[args.backbone]
        if 'adapter' in kind.lower():
            model = ctor(patch_size=args.patch_size, adapter_ffn_layernorm_option="none", adapter_ffn_init_option="lora",
                         adapter_ffn_scalar=str(args.adapter_ffn_scalar), adapter_ffn_num=args.ffn_num,
                         adapter_d_model=width)
        else:
            model = ctor(patch_size=args.patch_size, num_classes=0, use_adapter=False)
        num_feats = width
    for p in model.parameters():
        p.requires_grad = False
    model.configure(getattr(args, 'precision', 'fp32'))
    return model, num_feats


def _load_model_weights(args, embedder):
    """Checkpoint -> embedder by POSITION (zip over both key lists), exactly like the reference (compute_feats.py:477-480):
    DINO checkpoints carry a 'teacher' dict whose keys are prefixed differently from IClassifier's."""
    if not getattr(args, 'weights', None):
        print('Didnt load any weights for the embedder!')
        return None
    ckpt = torch.load(args.weights, map_location='cpu')
    if 'DINO' in args.embedder:
        weights = ckpt['teacher']
    elif 'MAE' in args.embedder:
        weights = ckpt['model']
    else:
        print('Didnt load any weights for the embedder!')
        return None
    new_state = OrderedDict()
    for (_, loaded_val), (init_key, _) in zip(weights.items(), embedder.state_dict().items()):
        new_state[init_key] = loaded_val
    msg = embedder.load_state_dict(new_state, strict=False)
    return msg


def get_embedder(args, backbone, num_feats):
    embedder = vit.IClassifier(backbone, num_feats, output_class=args.num_classes).to(device)
    _load_model_weights(args, embedder)
    return embedder, None


def write_bag_csv(path, feats, labels=None, positions=None, camelyon16=False, sidecar=False):
    """One CSV per slide: D feature columns named 0..D-1 [+ label, position], '%.4f' (reference compute_feats.py:256-266).

    ``sidecar=True`` also writes ``<path>.npz`` next to it: the SAME table as the loader would parse it from the text (the
    CSV just written is read back once, so the float32 values are bit-identical to what ``utils.get_bag_feats`` gets from
    the CSV -- '%.4f' rounding included).  A 30k x 768 slide is ~170 MB of text and seconds of parsing per epoch start;
    the sidecar is 92 MB and loads at memory speed.  The CSV stays the interchange format."""
    import pandas as pd
    df = pd.DataFrame(np.asarray(feats, dtype=np.float32))
    if camelyon16:
        df['label'] = labels if labels is not None else np.nan
        df['position'] = positions if positions is not None else None
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    df.to_csv(path, index=False, float_format='%.4f')
    if sidecar:
        write_bag_sidecar(path)


def write_bag_sidecar(csv_path):
    """Binary twin of an existing feature CSV (see write_bag_csv).  Returns the sidecar path."""
    import pandas as pd
    df = pd.read_csv(csv_path)
    has = 'position' in df and 'label' in df
    feats = (df.drop(columns=['label', 'position']) if has else df).to_numpy().astype('float32')
    out = csv_path + '.npz'
    tmp = out + '.tmp.npz'
    if has:
        np.savez(tmp, feats=feats, label=df['label'].to_numpy(), position=np.asarray(list(df['position']), dtype=object))
    else:
        np.savez(tmp, feats=feats)
    os.replace(tmp, out)
    return out


def compute_feats(args, bags_list, embedder, save_path, patch_labels_dict=None):
    """Per slide directory: batches of tiles -> embedder -> CSV (reference compute_feats.py:200-266).

    Features stay on the GPU until the slide is finished (one device->host copy per slide, not per batch)."""
    embedder.eval()
    if getattr(args, 'tune_gemms', 0):
        # library-GEMM selections for the extractor's skinny-K shapes (+13 % img/s at batch 512); shapes of other batch sizes
        # are tuned online on first use and recorded (see snuffy_amd/gemm_tuning.py)
        from .gemm_tuning import use_pretuned_gemms
        use_pretuned_gemms(tune_missing=True)
    for bag_dir in bags_list:
        patches = sorted(glob.glob(os.path.join(bag_dir, '*.jpg')) + glob.glob(os.path.join(bag_dir, '*.jpeg')))
        loader, _ = bag_dataset(args, patches, patch_labels_dict)
        feats, labels, positions = [], [], []
        with torch.no_grad():
            on_device = device_preprocess(args)
            for batch in loader:
                if on_device:   # uint8 tiles up, one kernel: Resize(224) + / 255 (+ ImageNet normalisation)
                    from .tiles import preprocess_tiles
                    u8 = batch['input'].to(device, non_blocking=True)
                    norm = getattr(args, 'transform', 0) == 1
                    fe = getattr(embedder, 'feature_extractor', None)
                    if (getattr(fe, 'precision', None) == 'bf16' and hasattr(fe, 'forward_cols')
                            and min(u8.shape[1], u8.shape[2]) >= 224 and u8.shape[1] == u8.shape[2]):
                        # bf16 extractor: the kernel writes the patch-embedding GEMM operand directly (no fp32 image, no patchify)
                        cols = preprocess_tiles(u8, 224, normalize=norm, want="cols", patch=fe.patch_embed.patch_size)
                        feats_b = fe.forward_cols(cols, u8.shape[0], 224, 224)
                        f = feats_b.view(feats_b.shape[0], -1)
                        feats.append(f)
                        labels.extend(np.atleast_1d(batch['label'].squeeze().tolist()).tolist())
                        positions.extend(batch['position'])
                        continue
                    x = preprocess_tiles(u8, 224, normalize=norm)
                else:
                    x = batch['input'].float().to(device, non_blocking=True)
                f, _ = embedder(x)
                feats.append(f)
                labels.extend(np.atleast_1d(batch['label'].squeeze().tolist()).tolist())
                positions.extend(batch['position'])
        if not feats:
            print('No valid patch extracted from: ' + bag_dir)
            continue
        split_name, class_name, bag_name = bag_dir.rstrip(os.path.sep).split(os.path.sep)[-3:]
        has = patch_labels_dict is not None
        write_bag_csv(os.path.join(save_path, split_name, class_name, bag_name + '.csv'), torch.cat(feats).cpu().numpy(),
                      labels if has else None, positions if has else None, camelyon16=(args.dataset == 'camelyon16'),
                      sidecar=bool(getattr(args, 'binary_sidecar', 0)))
