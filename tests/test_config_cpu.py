"""CPU: the config composer understands a Hydra tree shaped like the reference's (primary defaults list, config groups,
`# @package _global_` overlays carrying `override /group: option`, command-line grammar).  The tree is written by the test."""
import os

import pytest

from miphei_vit_amd.config import ConfigCompositionError, compose

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TREE = {
    "config.yaml": """
defaults:
  - _self_
  - data: siteA
  - train: cell
  - model: unet

data:
  stats: stats.json
  names: []
train:
  epochs: 20
  gan_train: false
  callbacks:
    ckpt: {mode: min, monitor: loss}
""",
    "data/siteA.yaml": """# @package _global_
data:
  names: [a, b, c]
train:
  epochs: 30
""",
    "data/siteB.yaml": """# @package _global_
data:
  names: [x]
""",
    "train/cell.yaml": """gan_train: true
use_cell_metrics: true
callbacks:
  ckpt: {mode: max, monitor: auc}
""",
    "train/structural.yaml": """use_cell_metrics: false
lr_ref: ${train.epochs}
""",
    "model/unet.yaml": """model_name: unet
dropout: 0.1
encoder:
  encoder_name: mobilenet_v2
  encoder_weights: imagenet
""",
    "recipes/vit.yaml": """# @package _global_
defaults:
  - override /train: structural

train:
  epochs: 15
  gan_train: false
model:
  model_name: myvitmatte
  encoder:
    encoder_name: hoptimus0
    encoder_weights: null
""",
    "recipes/bad.yaml": """# @package _global_
defaults:
  - override /optimizer: adam
x: 1
""",
    "recipes/nested.yaml": """# @package _global_
defaults:
  - /data: siteB
  - _self_
data:
  extra: 1
""",
    "recipes/scoped.yaml": """# @package model.encoder
depth: 40
""",
}


@pytest.fixture()
def tree(tmp_path):
    for rel, text in TREE.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    return str(tmp_path)


def test_primary_defaults_list_and_packages(tree):
    cfg = compose(tree)
    # _self_ first: the groups are merged OVER the primary config
    assert cfg.data.names == ["a", "b", "c"] and cfg.train.epochs == 30          # @package _global_ file
    assert cfg.train.gan_train is True and cfg.train.use_cell_metrics is True      # train/cell.yaml lands under `train`
    assert cfg.train.callbacks.ckpt == {"mode": "max", "monitor": "auc"}
    assert cfg.model.model_name == "unet" and cfg.model.encoder.encoder_name == "mobilenet_v2"
    assert cfg.data.stats == "stats.json"


def test_global_overlay_with_override_of_a_group(tree):
    cfg = compose(tree, ["+recipes=vit"])
    assert cfg.train.use_cell_metrics is False            # `override /train: structural` re-selected the group ...
    assert "callbacks" in cfg.train and cfg.train.callbacks.ckpt.mode == "min"   # ... so cell.yaml was never merged
    assert cfg.train.epochs == 15 and cfg.train.gan_train is False                # overlay merged last, at the root
    assert cfg.model.model_name == "myvitmatte" and cfg.model.encoder.encoder_weights is None
    assert cfg.model.dropout == 0.1                      # untouched keys of model/unet.yaml survive
    assert cfg.train.lr_ref == 15                        # ${train.epochs} resolved after composition
    assert "defaults" not in cfg and "recipes" not in cfg


def test_command_line_grammar(tree):
    cfg = compose(tree, ["train=structural", "data=siteB", "train.epochs=3", "+train.max_steps=7", "++model.dropout=0.0",
                         "++brand.new.key=[1,2]", "~model.encoder.encoder_weights"])
    assert cfg.data.names == ["x"] and cfg.train.use_cell_metrics is False
    assert cfg.train.epochs == 3 and cfg.train.max_steps == 7 and cfg.model.dropout == 0.0
    assert cfg.brand.new.key == [1, 2] and "encoder_weights" not in cfg.model.encoder
    with pytest.raises(ConfigCompositionError, match="Could not override 'train.nope'"):
        compose(tree, ["train.nope=1"])
    with pytest.raises(ConfigCompositionError, match="already at 'train.epochs'"):
        compose(tree, ["+train.epochs=1"])
    with pytest.raises(ConfigCompositionError, match="Could not override 'optimizer'"):
        compose(tree, ["+recipes=bad"])
    with pytest.raises(ConfigCompositionError, match="Could not find 'train/none'"):
        compose(tree, ["train=none"])
    with pytest.raises(ConfigCompositionError, match="Could not override 'recipes'"):
        compose(tree, ["recipes=vit"])                   # not in the defaults list: needs the + form


def test_nested_defaults_and_explicit_package(tree):
    cfg = compose(tree, ["+recipes=nested"])
    assert cfg.data.names == ["x"] and cfg.data.extra == 1
    cfg = compose(tree, ["+recipes=scoped"])
    assert cfg.model.encoder.depth == 40 and cfg.model.encoder.encoder_name == "mobilenet_v2"


def test_shipped_tree_composes_like_the_reference_command_lines():
    d = os.path.join(ROOT, "configs")
    cfg = compose(d, ["+default_configs=miphei-vit", "++train.epochs=100", "train.batch_size=8",
                      "++model.encoder.encoder_weights=null"])
    assert cfg.model.model_name == "myvitmatte" and cfg.model.encoder.encoder_name == "hoptimus0"
    assert cfg.train.epochs == 100 and cfg.train.batch_size == 8 and cfg.train.gan_train is False
    assert cfg.train.losses.lambda_factor == 50 and len(cfg.data.targ_channel_names) == 16
    assert cfg.model.encoder.encoder_weights is None and cfg.get_path("model.encoder.pretrained") is False
    assert cfg.train.use_cell_metrics is True             # train/cell.yaml, as in the reference recipe
    tiny = compose(d, ["+default_configs=tiny"])
    assert tiny.model.encoder.encoder_name == "tiny" and len(tiny.data.targ_channel_names) == 3
    unetr = compose(d, ["+default_configs=unetr"])
    assert unetr.model.model_name == "unet_lora" and unetr.model.dropout == 0.1 and unetr.train.batch_size == 8
    base = compose(d)                                     # bare `python run.py`: smp-UNet + GAN recipe -> outside the path
    assert base.model.model_name == "unet" and base.train.gan_train is True


def test_train_precision_is_checked_not_ignored():
    """cfg.train.precision (reference: pl.Trainer(precision=...), src/train.py:205-207): bf16-mixed is the path's one mode, the
    reference's shipped "16-mixed" is accepted with a warning, anything else raises."""
    import warnings
    from miphei_vit_amd.config import check_precision
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert check_precision("bf16-mixed") == "bf16-mixed"
    with pytest.warns(UserWarning, match="bf16-mixed"):
        assert check_precision("16-mixed") == "bf16-mixed"
    for bad in ("32-true", "64-true", "16-true", "bf16-true"):
        with pytest.raises(NotImplementedError):
            check_precision(bad)
