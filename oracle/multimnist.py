"""Oracle (test infrastructure): the recurrent text stacks of the MultiMNIST MVAE --
multimnist/model.py:145-235 -- restated on plain torch CPU ops with the GRU arithmetic written out
and the inter-layer dropout masks as explicit inputs.

    TextEncoder   multimnist/model.py:145-179   Embedding(12, 200) -> bidirectional 1-layer GRU over the
                  4 characters -> output at the LAST position (= forward state after 4 steps | backward
                  state after ONE step, on the last character) -> directions summed -> Linear(200, 2D)
    TextDecoder   multimnist/model.py:182-228   z2h(z) as the initial state of both GRU layers; 4 steps of
                  swish(Embedding(c_in)) | z -> 2-layer GRU (Dropout(0.1) between the layers in training
                  mode) -> | z -> Linear(200 + D, 12); c_in of the next step = arg-max of the logits
    constants     multimnist/utils.py:12-19     max_length 4, n_characters 10 + 2, SOS 10, FILL 11
    text loss     multimnist/train.py:47-58,100-117   cross_entropy summed over classes and the 4 digits

nn.GRU's cell (torch's documented gate order r | z | n):
    r = s(W_ir x + b_ir + W_hr h + b_hr); z likewise; n = tanh(W_in x + b_in + r * (W_hn h + b_hn));
    h' = (1 - z) * n + z * h.
``nn.GRU(200, 200, 1, dropout=0.1)`` never drops anything (dropout sits BETWEEN layers);
``nn.GRU(200 + D, 200, 2, dropout=0.1)`` multiplies layer 0's output by ``bernoulli(0.9) / 0.9`` in
training mode, one [B, 200] draw per decoder step from the global generator (verified against the
imported reference by tests/golden/make_multimnist_golden.py).  Parameter names equal the reference's
``state_dict`` keys (``gru.weight_ih_l0`` ... ``gru.bias_hh_l0_reverse``)."""
import torch
import torch.nn as nn

from .functional import cross_entropy, swish

MAX_LENGTH = 4
N_CHARACTERS = 12
SOS, FILL = 10, 11
KEEP = 0.9


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    H = h.shape[1]
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def _gru_params(gru, layer, reverse=False):
    sfx = '_l%d%s' % (layer, '_reverse' if reverse else '')
    return (getattr(gru, 'weight_ih' + sfx), getattr(gru, 'weight_hh' + sfx),
            getattr(gru, 'bias_ih' + sfx), getattr(gru, 'bias_hh' + sfx))


class TextEncoder(nn.Module):
    def __init__(self, n_latents, n_characters=N_CHARACTERS, n_hiddens=200, bidirectional=True):
        super().__init__()
        self.embed = nn.Embedding(n_characters, n_hiddens)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')        # "dropout expects num_layers > 1": the reference asks for it too
            self.gru = nn.GRU(n_hiddens, n_hiddens, 1, dropout=0.1, bidirectional=bidirectional)   # parameter holder
        self.h2p = nn.Linear(n_hiddens, n_latents * 2)
        self.n_latents, self.n_hiddens, self.bidirectional = n_latents, n_hiddens, bidirectional

    def forward(self, x):
        B, L = x.shape
        e = self.embed(x)                                          # [B, L, H]
        h = e.new_zeros(B, self.n_hiddens)
        for t in range(L):                                         # forward direction: all L steps
            h = gru_cell(e[:, t], h, *_gru_params(self.gru, 0))
        out = h
        if self.bidirectional:                                     # backward direction at the last position: ONE step
            hb = gru_cell(e[:, L - 1], e.new_zeros(B, self.n_hiddens), *_gru_params(self.gru, 0, True))
            out = h + hb
        p = self.h2p(out)
        return p[:, :self.n_latents], p[:, self.n_latents:]


class TextDecoder(nn.Module):
    def __init__(self, n_latents, n_characters=N_CHARACTERS, n_hiddens=200):
        super().__init__()
        self.embed = nn.Embedding(n_characters, n_hiddens)
        self.z2h = nn.Linear(n_latents, n_hiddens)
        self.gru = nn.GRU(n_hiddens + n_latents, n_hiddens, 2, dropout=0.1)                         # parameter holder
        self.h2o = nn.Linear(n_hiddens + n_latents, n_characters)
        self.n_latents, self.n_characters, self.n_hiddens = n_latents, n_characters, n_hiddens

    def forward(self, z, dropout_masks=None):
        """``dropout_masks``: MAX_LENGTH tensors [B, 200] in {0, 1} (training mode; drawn here in the
        reference's order when omitted).  Returns (words [B, 4, 12] logits, the fed-back characters [4, B])."""
        B = z.shape[0]
        if self.training and dropout_masks is None:
            dropout_masks = draw_decoder_masks(B, self.n_hiddens)
        c_in = torch.full((B,), SOS, dtype=torch.long)
        h0 = h1 = self.z2h(z)
        words, fed = [], []
        for i in range(MAX_LENGTH):
            fed.append(c_in)
            x = torch.cat((swish(self.embed(c_in)), z), dim=1)
            h0 = gru_cell(x, h0, *_gru_params(self.gru, 0))
            d = h0 * dropout_masks[i] / KEEP if self.training else h0
            h1 = gru_cell(d, h1, *_gru_params(self.gru, 1))
            c_out = self.h2o(torch.cat((h1, z), dim=1))
            words.append(c_out)
            c_in = torch.max(torch.log_softmax(c_out, dim=1), dim=1)[1]
        return torch.stack(words, dim=1), torch.stack(fed)


def draw_decoder_masks(batch, n_hiddens=200, generator=None):
    """One Bernoulli(0.9) draw of [B, 200] per decoder step, in step order (the reference's at::dropout
    inside nn.GRU draws ``empty_like(layer-0 output).bernoulli_(0.9)``)."""
    return [torch.empty(batch, n_hiddens).bernoulli_(KEEP, generator=generator) for _ in range(MAX_LENGTH)]


def text_loss_rows(recon_text, text):
    """The text half of multimnist/train.py:47-58: cross_entropy over the 12 classes, summed over the
    classes and over the 4 digits -> [B]."""
    B, L, K = recon_text.shape
    ce = cross_entropy(recon_text.reshape(-1, K), text.reshape(-1)).sum(dim=1)
    return ce.view(B, L).sum(dim=1)


def synthetic_text(batch, seed):
    """Random MultiMNIST labels: 0-4 digits, FILL-padded (multimnist/utils.py:22-31 char_tensor)."""
    g = torch.Generator().manual_seed(seed)
    digits = torch.randint(0, 10, (batch, MAX_LENGTH), generator=g)
    n = torch.randint(0, MAX_LENGTH + 1, (batch,), generator=g)
    pos = torch.arange(MAX_LENGTH).unsqueeze(0)
    return torch.where(pos < n.unsqueeze(1), digits, torch.full_like(digits, FILL))
