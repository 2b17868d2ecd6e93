"""
Decoders -- SDF and colour decoders with the API and state_dict layout of reference src/networks/decoders.py
(class Decoders, :24-205), computing on the fused MFMA MLP kernel (csrc/mlp.hip) and the HIP hash grid.

  cfg['grid']['tcnn_network'] True : self.sdf_decoder / self.color_decoder are FusedMLP modules with tcnn's flat
        `params` (keys sdf_decoder.params, color_decoder.params), no bias, n_hidden_layers = n_blocks-1   (:49-70)
  cfg['grid']['tcnn_network'] False: nn.Linear stacks with bias (keys linears.*, c_linears.*, output_linear.*,
        c_output_linear.*) exactly as the reference (:72-84); their weights are packed into the kernel's flat
        layout on every call (a few hundred floats) so autograd returns per-layer gradients.
"""
import torch
import torch.nn as nn

from .hashgrid import HashGridEncoding
from .network import FusedMLP, fused_mlp, make_mlp_desc


class Decoders(nn.Module):
    def __init__(self, cfg, c_dim=32, hidden_size=16, truncation=0.08, n_blocks=2, learnable_beta=True):
        super().__init__()
        self.c_dim = c_dim
        self.cfg = cfg
        self.truncation = truncation
        self.n_blocks = n_blocks
        self.hidden_size = hidden_size
        self.tcnn_network = self.cfg['grid']['tcnn_network']
        # extension over the reference's yaml: MFMA operand type of the decoder MLPs ("fp32" | "bf16")
        self.mlp_precision = self.cfg.get('model', {}).get('mlp_precision', 'fp32') if hasattr(self.cfg, 'get') else 'fp32'
        if cfg['grid_mode'] != 'hash_grid':
            raise ValueError("Decoders: only grid_mode == 'hash_grid' exists in Uni-SLAM (decoders.py:116-120)")
        input_channels = c_dim
        if self.tcnn_network:
            mk = lambda n_out, act: FusedMLP(input_channels, n_out, {
                "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act,
                "n_neurons": hidden_size, "n_hidden_layers": n_blocks - 1, "precision": self.mlp_precision})
            self.sdf_decoder = mk(1, "Tanh")
            self.color_decoder = mk(3, "Sigmoid")
        else:
            self.linears = nn.ModuleList([nn.Linear(input_channels, hidden_size)] +
                                         [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
            self.c_linears = nn.ModuleList([nn.Linear(input_channels, hidden_size)] +
                                           [nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
            self.output_linear = nn.Linear(hidden_size, 1)
            self.c_output_linear = nn.Linear(hidden_size, 3)
            if n_blocks not in (1, 2):
                raise ValueError("Decoders: the fused kernel covers n_blocks in {1, 2}")
            self._desc_sdf, self._desc_rgb = self.mlp_descs()
        if learnable_beta:
            self.beta = nn.Parameter(10 * torch.ones(1))
        else:
            self.beta = 10

    # descriptors are ctypes objects: rebuild them after pickling / deepcopy (Tracker.py:106, UNISLAM.py:295-298)
    def __getstate__(self):
        s = self.__dict__.copy()
        s.pop("_desc_sdf", None); s.pop("_desc_rgb", None)
        return s

    def __setstate__(self, s):
        self.__dict__.update(s)
        self.__dict__.setdefault("mlp_precision", "fp32")
        if not self.tcnn_network:
            self._desc_sdf, self._desc_rgb = self.mlp_descs()

    def mlp_descs(self):
        """(sdf, colour) us_mlp_desc of the two decoders"""
        if self.tcnn_network:
            return self.sdf_decoder.desc, self.color_decoder.desc
        return (make_mlp_desc(self.c_dim, self.hidden_size, self.n_blocks, 1, "tanh", True, self.mlp_precision),
                make_mlp_desc(self.c_dim, self.hidden_size, self.n_blocks, 3, "sigmoid", True, self.mlp_precision))

    def __deepcopy__(self, memo):
        new = Decoders(self.cfg, self.c_dim, self.hidden_size, self.truncation, self.n_blocks,
                       isinstance(self.beta, nn.Parameter))
        new.to(next(self.parameters()).device)
        new.load_state_dict(self.state_dict())
        for (_, a), (_, b) in zip(self.named_parameters(), new.named_parameters()):
            b.requires_grad_(a.requires_grad)
        if hasattr(self, "bound"):
            new.bound = self.bound
        return new

    @staticmethod
    def pack_linear_params(hidden, out):
        """flat fp32 vector in us_mlp_desc layout: weights (last matrix zero-padded to 16 rows), then biases"""
        w = [l.weight.reshape(-1) for l in hidden]
        pad_w = out.weight.new_zeros((16 - out.weight.shape[0], out.weight.shape[1]))
        w.append(torch.cat([out.weight, pad_w], 0).reshape(-1))
        b = [l.bias for l in hidden]
        b.append(torch.cat([out.bias, out.bias.new_zeros(16 - out.bias.shape[0])]))
        return torch.cat(w + b)

    def sample_hash_grid_feature(self, p_nor, hash_grids_xyz):
        """decoders.py:91-105: clamp to [0,1], then encode (the clamp is folded into the HIP kernel)."""
        enc = hash_grids_xyz[0]
        if isinstance(enc, HashGridEncoding):
            return enc(p_nor, clamp=True)
        return enc(torch.clamp(p_nor, min=0, max=1))

    def get_raw_sdf(self, p_nor, scene_rep):
        """decoders.py:107-130"""
        hash_grids_xyz, c_hash_grids_xyz = scene_rep[0], scene_rep[1]
        h = self.sample_hash_grid_feature(p_nor, hash_grids_xyz)
        if self.tcnn_network:
            return self.sdf_decoder(h).squeeze()
        return fused_mlp(h, self.pack_linear_params(self.linears, self.output_linear), self._desc_sdf).squeeze()

    def get_raw_rgb(self, p_nor, scene_rep):
        """decoders.py:132-155"""
        hash_grids_xyz, c_hash_grids_xyz = scene_rep[0], scene_rep[1]
        h = self.sample_hash_grid_feature(p_nor, c_hash_grids_xyz)
        if self.tcnn_network:
            return self.color_decoder(h)
        return fused_mlp(h, self.pack_linear_params(self.c_linears, self.c_output_linear), self._desc_rgb)

    def forward(self, p, scene_rep):
        """decoders.py:182-205: p [..., 3] (already normalised to [0,1]) -> raw [..., 4] = (rgb, sdf)"""
        p_shape = p.shape
        p_nor = p.reshape(-1, 3)
        sdf = self.get_raw_sdf(p_nor, scene_rep)
        rgb = self.get_raw_rgb(p_nor, scene_rep)
        raw = torch.cat([rgb, sdf.reshape(-1, 1)], dim=-1)
        return raw.reshape(*p_shape[:-1], -1)


def get_model(cfg):
    """reference src/networks/config.py:20-27"""
    return Decoders(cfg, c_dim=cfg['model']['c_dim'], truncation=cfg['model']['truncation'],
                    learnable_beta=cfg['rendering']['learnable_beta'])
