"""Oracle (test infrastructure): the reference's loss / latent arithmetic, restated.

Every function cites the reference lines it follows (paths relative to the
reference repository root).  Plain torch CPU fp32 ops only.
"""
import torch
import torch.nn.functional as F

POE_EPS = 1e-8


def swish(x):
    """x * sigmoid(x) -- mnist/model.py:166-169 (same text in every experiment)."""
    return x * torch.sigmoid(x)


def poe(mu, logvar, variant):
    """Product of Gaussian experts over dim 0 of ``[M, B, D]`` stacks.

    variant 'A' -- mnist/model.py:156-163, fashionmnist/model.py:175-182:
        var = exp(lv) + eps; T = 1/(var + eps); lv_out = log(1/sum(T) + eps)
    variant 'B' -- celeba/model.py:200-207, celeba19/model.py:219-226:
        var = exp(lv) + eps; T = 1/var;         lv_out = log(1/sum(T))
    """
    eps = POE_EPS
    var = torch.exp(logvar) + eps
    if variant == 'A':
        T = 1. / (var + eps)
    elif variant == 'B':
        T = 1. / var
    else:
        raise ValueError(variant)
    pd_mu = torch.sum(mu * T, dim=0) / torch.sum(T, dim=0)
    pd_var = 1. / torch.sum(T, dim=0)
    pd_logvar = torch.log(pd_var + eps) if variant == 'A' else torch.log(pd_var)
    return pd_mu, pd_logvar


def prior_expert(batch, n_latents):
    """N(0, 1) expert, ``[1, B, D]`` zeros for both mu and logvar --
    mnist/model.py:172-185 (celeba uses log(ones) == zeros, celeba/model.py:216-229)."""
    z = torch.zeros(1, batch, n_latents)
    return z, z.clone()


def poe_with_prior(expert_mus, expert_logvars, variant):
    """MVAE.infer's stacking: prior first, then the present experts in order --
    mnist/model.py:46-64, celeba19/model.py:63-89."""
    b, d = expert_mus[0].shape
    mu, lv = prior_expert(b, d)
    mu = torch.cat([mu] + [m.unsqueeze(0) for m in expert_mus], dim=0)
    lv = torch.cat([lv] + [v.unsqueeze(0) for v in expert_logvars], dim=0)
    return poe(mu, lv, variant)


def reparametrize(mu, logvar, eps):
    """train mode: eps * exp(0.5 * logvar) + mu; eval (eps None): mu --
    mnist/model.py:29-35."""
    if eps is None:
        return mu
    std = logvar.mul(0.5).exp()
    return eps.mul(std).add(mu)


def binary_cross_entropy_with_logits(input, target):
    """Elementwise: clamp(x, 0) - x t + log(1 + exp(-|x|)) -- mnist/train.py:62-74.
    Raises ValueError on a size mismatch, like the reference (:69-71)."""
    if not (target.size() == input.size()):
        raise ValueError("Target size ({}) must be the same as input size ({})".format(
            target.size(), input.size()))
    return (torch.clamp(input, 0) - input * target
            + torch.log(1 + torch.exp(-torch.abs(input))))


def cross_entropy(input, target, eps=1e-6):
    """-onehot(target) * log_softmax(input + eps) as ``[B, K]`` -- mnist/train.py:77-94."""
    if not (target.size(0) == input.size(0)):
        raise ValueError(
            "Target size ({}) must be the same as input size ({})".format(
                target.size(0), input.size(0)))
    log_input = F.log_softmax(input + eps, dim=1)
    y_onehot = torch.zeros_like(log_input).scatter(1, target.unsqueeze(1), 1)
    return -(y_onehot * log_input)


def kl_rows(mu, logvar):
    """-0.5 * sum_d(1 + lv - mu^2 - exp(lv)) -- mnist/train.py:56."""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1)


def elbo_loss_label(recon_image, image, recon_text, text, mu, logvar,
                    lambda_image=1.0, lambda_text=1.0, annealing_factor=1):
    """Bimodal ELBO with a categorical label -- mnist/train.py:20-59
    (fashionmnist/train.py is the same text)."""
    image_bce, text_bce = 0, 0
    if recon_image is not None and image is not None:
        n = image[0].numel()
        image_bce = torch.sum(binary_cross_entropy_with_logits(
            recon_image.reshape(-1, n), image.reshape(-1, n)), dim=1)
    if recon_text is not None and text is not None:
        text_bce = torch.sum(cross_entropy(recon_text, text), dim=1)
    KLD = kl_rows(mu, logvar)
    return torch.mean(lambda_image * image_bce + lambda_text * text_bce
                      + annealing_factor * KLD)


def elbo_loss_attrs(recon_image, image, recon_attrs, attrs, mu, logvar,
                    lambda_image=1.0, lambda_attrs=1.0, annealing_factor=1):
    """Bimodal ELBO with 18 Bernoulli attributes, summed column by column --
    celeba/train.py:22-65 (column loop :54-58)."""
    image_bce, attrs_bce = 0, 0
    if recon_image is not None and image is not None:
        image_bce = torch.sum(binary_cross_entropy_with_logits(
            recon_image.reshape(-1, 3 * 64 * 64), image.reshape(-1, 3 * 64 * 64)), dim=1)
    if recon_attrs is not None and attrs is not None:
        for i in range(attrs.size(1)):
            attrs_bce = attrs_bce + binary_cross_entropy_with_logits(
                recon_attrs[:, i], attrs[:, i])
    KLD = kl_rows(mu, logvar)
    return torch.mean(lambda_image * image_bce + lambda_attrs * attrs_bce
                      + annealing_factor * KLD)


def elbo_loss_multi(recon, data, mu, logvar, lambda_image=1.0,
                    lambda_attrs=1.0, annealing_factor=1.):
    """N-modal ELBO over lists -- celeba19/train.py:26-60.  A list entry with
    more than one dim is an image (:52-55), otherwise a single attribute (:56-57)."""
    assert len(recon) == len(data), "must supply ground truth for every modality."
    batch_size = mu.size(0)
    BCE = 0
    for ix in range(len(recon)):
        if recon[ix].dim() > 1:
            recon_ix = recon[ix].reshape(batch_size, -1)
            data_ix = data[ix].reshape(batch_size, -1)
            BCE = BCE + lambda_image * torch.sum(
                binary_cross_entropy_with_logits(recon_ix, data_ix), dim=1)
        else:
            BCE = BCE + lambda_attrs * binary_cross_entropy_with_logits(recon[ix], data[ix])
    KLD = kl_rows(mu, logvar)
    return torch.mean(BCE + annealing_factor * KLD)
