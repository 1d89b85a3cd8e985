"""a4 — multi-view voxel aggregation: 3-D ConvGRU over the view sequence.

Mirror of the reference's models/fusion.py (ConvGRUCell_3D :7-35, ConvGRU_3D :39-95): same class
names, constructor arguments, sub-module names and therefore state_dict keys
(`cells.0.conv_gate.weight [256,256,3,3,3]`, `cells.0.out_gate.*`, `fusion_norm.*`,
`fusion_conv.{0,1,3,4}.*`). The dead bookkeeping of the reference forward (layer_output_list,
last_state_list, torch.stack of every h) is not reproduced — only the returned value is.
"""
import torch
import torch.nn as nn


class ConvGRUCell_3D(nn.Module):
    """models/fusion.py:7-35. Gate split order is (update, reset) (:30)."""

    def __init__(self, config, input_size, hidden_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.conv_gate = nn.Conv3d(input_size + hidden_size, hidden_size * 2, 3, padding=1)
        self.out_gate = nn.Conv3d(input_size + hidden_size, hidden_size, 3, padding=1)

    def forward(self, x, prev_state=None):
        b, c, d, h, w = x.shape
        if prev_state is None:
            prev_state = torch.zeros([b, self.hidden_size, d, h, w], dtype=x.dtype, device=x.device)
        gates = self.conv_gate(torch.cat([x, prev_state], dim=1))
        update, reset = torch.split(gates, self.hidden_size, dim=1)
        update, reset = torch.sigmoid(update), torch.sigmoid(reset)
        out_inputs = torch.tanh(self.out_gate(torch.cat([x, prev_state * reset], dim=1)))
        return prev_state * (1 - update) + out_inputs * update


class ConvGRU_3D(nn.Module):
    """models/fusion.py:39-95."""

    def __init__(self, config, n_layers=1, input_size=16, hidden_size=16):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.cells = nn.ModuleList([
            ConvGRUCell_3D(config, input_size if i == 0 else hidden_size, hidden_size) for i in range(n_layers)])
        self.fusion_norm = nn.BatchNorm3d(hidden_size)
        self.fusion_conv = nn.Sequential(
            nn.Conv3d(input_size, input_size, 3, padding=1),
            nn.BatchNorm3d(input_size),
            nn.LeakyReLU(inplace=True),
            nn.Conv3d(input_size, input_size, 3, padding=1),
            nn.BatchNorm3d(input_size),
            nn.LeakyReLU(inplace=True),
        )

    def forward(self, x, hidden=None):
        """x [b,t,c,d,h,w] -> fusion_norm(h_T) [b,c',d,h,w]"""
        seq_len = x.shape[1]
        if not hidden:
            hidden = [None] * self.n_layers
        cur = x
        h = None
        for layer_idx in range(self.n_layers):
            h = hidden[layer_idx]
            outs = []
            for t in range(seq_len):
                h = self.cells[layer_idx](cur[:, t], h)
                if layer_idx + 1 < self.n_layers:
                    outs.append(h)
            if layer_idx + 1 < self.n_layers:
                cur = torch.stack(outs, dim=1)
        return self.fusion_norm(h)
