"""Post-norm feed-forward block: x -> LN(x + W2 relu(W1 x)) (reference models/ffn.py)."""
import torch.nn as nn

from ..functions.clip_ops import add_layer_norm


class FFN(nn.Module):
    def __init__(self, d_model, d_ffn, dropout: float):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = nn.ReLU(inplace=True)
        self.dropout1 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout2 = nn.Dropout(dropout)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt):
        hidden = self.dropout1(self.activation(self.linear1(tgt)))
        return add_layer_norm(tgt, self.dropout2(self.linear2(hidden)), self.norm)
