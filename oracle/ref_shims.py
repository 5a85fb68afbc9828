"""TEST INFRASTRUCTURE — golden-vector generation only, runs only in the build container.

Imports the UNMODIFIED reference Python modules from /root/reference under dependency stubs
(torch_geometric, torch_scatter, hydra are not installed).  Nothing in
`tests/ -m gpu`, `smoke()` or `bench.py` imports this file: /root/reference does not exist on the
GPU box.  The scripts that use it (`oracle/gen_golden.py`) write small data fixtures into
`tests/golden/`; no reference source text is copied anywhere.
"""
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "policies"))


def install():
    """Make `policies.*`, `modules.*`, `datasets.rl_waymo.dataset`, `utils.*` importable from REF."""
    import torch

    if "policies.autoregressive_policy" in sys.modules:
        return
    sys.path.insert(0, REF)  # must precede site-packages: HF `datasets` is installed

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class _Store(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    class HeteroData(dict):
        def __init__(self, d=None):
            super().__init__()
            for k, v in (d or {}).items():
                self[k] = _Store(v) if isinstance(v, dict) else v

        def cuda(self):
            return self

    class Dataset:
        def __init__(self, *a, **k):
            pass

        def __getitem__(self, i):
            return self.get(i)

    stub("torch_geometric")
    stub("torch_geometric.data", HeteroData=HeteroData, Dataset=Dataset, Data=object, Batch=object)
    stub("torch_geometric.data.storage", BaseStorage=object, EdgeStorage=object, NodeStorage=object)
    stub("torch_geometric.loader", DataLoader=object)
    stub("torch_scatter")
    stub("hydra", main=lambda **k: (lambda f: f))
    stub("cfgs")
    stub("cfgs.config", CONFIG_PATH="")
    for pkg in ("utils", "modules", "policies", "datasets", "datasets.rl_waymo"):
        m = types.ModuleType(pkg)
        m.__path__ = [f"{REF}/{pkg.replace('.', '/')}"]
        sys.modules[pkg] = m
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self


def build_reference_model(cfg, weights):
    """nn.Module wrapping the reference Encoder/Decoder exactly as models/ctrl_sim.py:41-45 does
    (the LightningModule itself needs pytorch_lightning, which is absent)."""
    install()
    import torch
    import torch.nn as nn
    from modules.encoder import Encoder
    from modules.decoder import Decoder

    class RefCtRLSim(nn.Module):
        def __init__(self):
            super().__init__()
            self.cfg = cfg
            self.encoder = Encoder(cfg)
            self.decoder = Decoder(cfg)

        def forward(self, data, eval=False):
            scene_enc = self.encoder(data, eval)
            return self.decoder(data, scene_enc, eval)

    m = RefCtRLSim()
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    # the only non-parameter state is nothing: causal_mask is a plain attribute
    assert not missing and not unexpected, (missing, unexpected)
    m.eval()
    return m


def build_reference_dataset(cfg):
    install()
    from datasets.rl_waymo.dataset import RLWaymoDataset

    return RLWaymoDataset(cfg, split_name="test", mode="eval")
