"""ReLU MLP with parameters under ``layers.{i}`` (state-dict layout of the reference's models/mlp.py)."""
import torch.nn as nn
from ..modules.linear import row_linear


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        last = self.num_layers - 1
        for i, layer in enumerate(self.layers):
            x = row_linear(x, layer.weight, layer.bias, relu=i < last)
        return x
