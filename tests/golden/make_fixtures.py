#!/usr/bin/env python3
"""Generate golden vectors from the reference's own archs/uPIT.py.  Run in the build
container only (needs /root/reference); the .npz files it writes are committed and are
the only thing that travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py

The reference module does not import on a modern stack for three ordinary reasons
(SURVEY.md 8c); its *source text* is loaded in memory with these compatibility shims:
  1. `numpy_type_map` (unused, removed from torch) dropped from the import at archs/uPIT.py:12
  2. collections.Mapping -> collections.abc.Mapping              (archs/uPIT.py:39)
  3. `.cuda()` made a no-op on Tensor / PackedSequence, SepDNN(-1) (archs/uPIT.py:160,164,179,210)
h0/c0 are captured by wrapping init_hidden, because the reference draws them with randn
for every batch (archs/uPIT.py:121-127).

Weights are NOT stored (13.4 M floats): both sides build the model under the same
torch.manual_seed on the same torch build, and the fixture carries per-parameter
checksums so a drifted init is detected rather than silently compared.
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np
import torch
from torch.nn.utils.rnn import PackedSequence

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_rsh():
    """archs/RSH.py with the same three shims as uPIT (numpy_type_map import, collections.Mapping, .cuda())."""
    load_reference_upit()                      # installs the shims and the plot stub
    src = open(os.path.join(REF, "archs/RSH.py")).read()
    src = src.replace("from torch.utils.data.dataloader import default_collate, numpy_type_map",
                      "from torch.utils.data.dataloader import default_collate")
    mod = types.ModuleType("ref_RSH")
    exec(compile(src, os.path.join(REF, "archs/RSH.py"), "exec"), mod.__dict__)
    return mod


def load_reference_upit():
    collections.Mapping = collections.abc.Mapping
    torch.Tensor.cuda = lambda self, *a, **k: self
    PackedSequence.cuda = lambda self, *a, **k: self
    plot_stub = types.ModuleType("plot")          # tools/plot.py needs matplotlib; never called here
    plot_stub.plot_spec = lambda *a, **k: None
    plot_stub.plot_loss = lambda *a, **k: None
    sys.modules["plot"] = plot_stub
    src = open(os.path.join(REF, "archs/uPIT.py")).read()
    src = src.replace("from torch.utils.data.dataloader import default_collate, numpy_type_map",
                      "from torch.utils.data.dataloader import default_collate")
    mod = types.ModuleType("ref_uPIT")
    exec(compile(src, os.path.join(REF, "archs/uPIT.py"), "exec"), mod.__dict__)
    return mod


def synth_batch(rng, lens, feat_dim, num_spk, with_names=False):
    """Magnitude-spectrogram-like positive features; mix is NOT the sum of sources on purpose
    (the loss must not rely on it)."""
    samples = []
    for i, T in enumerate(lens):
        d = {"mix": np.abs(rng.standard_normal((T, feat_dim))).astype(np.float32)}
        if with_names:
            d["name"] = "utt%02d.npz" % i
        else:
            for s in range(num_spk):
                d["source%d" % (s + 1)] = np.abs(rng.standard_normal((T, feat_dim))).astype(np.float32) * 0.7
        samples.append(d)
    return samples


def param_checksums(model):
    return {k: np.array([float(v.double().sum()), float(v.double().abs().sum())])
            for k, v in model.state_dict().items() if v.dtype.is_floating_point}


def capture_hidden(model):
    """Wrap init_hidden so the randn draws are recorded."""
    rec = {}
    orig = model.init_hidden

    def wrapped(batch_size):
        h = orig(batch_size)
        rec["h0"], rec["c0"] = h[0].clone(), h[1].clone()
        return h
    model.init_hidden = wrapped
    return rec


def rsh_samples(rng, spec, feat_dim, test=False):
    """spec: list of (T, num_spk).  'combo' = [mixture | ones] as TrainSet/TestSet build it (archs/RSH.py:104-106)."""
    out = []
    for i, (T, n) in enumerate(spec):
        mix = np.abs(rng.standard_normal((T, feat_dim))).astype(np.float32)
        d = {"combo": np.concatenate((mix, np.ones(mix.shape)), axis=1)}
        if test:
            d["name"] = "utt%02d.npz" % i
            d["num_spk"] = n
        else:
            for s in range(n):
                d["source%d" % (s + 1)] = (np.abs(rng.standard_normal((T, feat_dim))) * 0.6).astype(np.float32)
        out.append(d)
    return out


def main_rsh():
    m = load_reference_rsh()
    torch.set_num_threads(4)
    spec = [(9, 2), (7, 3), (11, 2), (8, 3), (7, 2)]
    # ---- training loss + grads over a mixed 2-/3-speaker batch
    torch.manual_seed(2024)
    model = m.SepDNN(-1)
    model.train()
    hid = []
    orig = model.init_hidden
    model.init_hidden = lambda b: hid.append(orig(b)) or hid[-1]
    rng = np.random.default_rng(2024)
    samples = rsh_samples(rng, spec, 257)
    batch = m.Collator("combo")(samples)
    loss, norm = m.compute_loss(model, 0, batch)
    loss.backward()
    fx = {"seed": np.array(2024), "spec": np.array(spec), "loss": loss.detach().numpy(), "norm": norm.detach().numpy(),
          "sub_batch_lens": np.array(batch.sub_batch_lens),
          "running_mean": model.bn.running_mean.numpy().copy(), "running_var": model.bn.running_var.numpy().copy(),
          "num_batches_tracked": model.bn.num_batches_tracked.numpy().copy()}
    for j, (h, c) in enumerate(hid):
        fx["h0_%d" % j], fx["c0_%d" % j] = h.numpy(), c.numpy()
    for i, d in enumerate(samples):
        for k, v in d.items():
            fx["sample%d_%s" % (i, k)] = v.astype(np.float32)
    for k, v in param_checksums(model).items():
        fx["wsum_" + k] = v
    for k, p in model.named_parameters():
        g = p.grad
        fx["gnorm_" + k] = np.array(float(g.double().norm()))
        flat = g.flatten()
        fx["gslice_" + k] = flat[:: max(1, flat.numel() // 64)][:64].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "ref_rsh_loss.npz"), **fx)

    # ---- compute_masks (eval mode, running stats, no relu in the attention update)
    torch.manual_seed(555)
    model = m.SepDNN(-1)
    with torch.no_grad():
        model.bn.running_mean.normal_(0.0, 0.05)
        model.bn.running_var.uniform_(0.5, 1.5)
    model.eval()
    hid = []
    orig = model.init_hidden
    model.init_hidden = lambda b: hid.append(orig(b)) or hid[-1]
    rng = np.random.default_rng(555)
    tspec = [(8, 2), (6, 3), (10, 2)]
    samples = rsh_samples(rng, tspec, 257, test=True)
    batch = m.Collator("combo")(samples)
    outdir = "/tmp/ref_rsh_masks_fixture"
    os.makedirs(outdir, exist_ok=True)
    with torch.no_grad():
        m.compute_masks(model, batch, outdir)
    fx = {"seed": np.array(555), "spec": np.array(tspec), "running_mean": model.bn.running_mean.numpy(),
          "running_var": model.bn.running_var.numpy()}
    for j, (h, c) in enumerate(hid):
        fx["h0_%d" % j], fx["c0_%d" % j] = h.numpy(), c.numpy()
    for i, d in enumerate(samples):
        fx["sample%d_combo" % i] = d["combo"].astype(np.float32)
        z = np.load(os.path.join(outdir, d["name"]))
        for k in z.files:
            fx["mask_%s_%s" % (d["name"], k)] = z[k]
    np.savez_compressed(os.path.join(HERE, "ref_rsh_masks.npz"), **fx)
    print("rsh fixtures written: loss", float(loss), "norm", float(norm))


def main():
    m = load_reference_upit()
    torch.set_num_threads(4)
    out = {}

    # ---- case 1: training loss + grads, S=2, ragged lengths (unsorted, with a tie) ----------
    for tag, num_spk, lens, seed in (("s2", 2, [9, 12, 7, 12], 1234), ("s3", 3, [6, 10, 8], 4321)):
        torch.manual_seed(seed)
        model = m.SepDNN(-1, num_spk=str(num_spk))
        model.train()
        rec = capture_hidden(model)
        rng = np.random.default_rng(seed)
        samples = synth_batch(rng, lens, 257, num_spk)
        coll = m.Collator("mix")
        batch = coll(samples)
        order = np.argsort(np.array(lens))[::-1]
        loss, norm = m.compute_loss(model, 0, batch)
        loss.backward()
        sums = param_checksums(model)              # before the extra forward below moves BN stats again
        run_mean, run_var = model.bn.running_mean.numpy().copy(), model.bn.running_var.numpy().copy()
        # recompute the pieces compute_loss keeps local, from the captured hidden state
        model.hidden = (rec["h0"], rec["c0"])
        with torch.no_grad():
            mask_out = model(batch["mix"])
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        fx = {
            "seed": np.array(seed), "num_spk": np.array(num_spk), "lens": np.array(lens),
            "order": order.copy(),
            "h0": rec["h0"].numpy(), "c0": rec["c0"].numpy(),
            "loss": loss.detach().numpy(), "norm": norm.detach().numpy(),
            "mask_out": mask_out.numpy(),
            "running_mean": run_mean, "running_var": run_var,
        }
        for i, d in enumerate(samples):
            for k, v in d.items():
                fx["sample%d_%s" % (i, k)] = v
        for k, v in sums.items():
            fx["wsum_" + k] = v
        for k, g in grads.items():
            fx["gnorm_" + k] = np.array(float(g.double().norm()))
            flat = g.flatten()
            fx["gslice_" + k] = flat[:: max(1, flat.numel() // 64)][:64].numpy().copy()
        np.savez_compressed(os.path.join(HERE, "ref_upit_loss_%s.npz" % tag), **fx)
        out[tag] = float(loss)

    # ---- case 2: three optimisation steps of steps/train_qsub.py:116-122 ------------------
    torch.manual_seed(77)
    model = m.SepDNN(-1)
    model.train()
    rec = capture_hidden(model)
    opt = torch.optim.Adam(model.parameters(), lr=0.001)
    rng = np.random.default_rng(77)
    fx = {"seed": np.array(77)}
    coll = m.Collator("mix")
    for step, lens in enumerate(([8, 5, 11], [10, 10, 4], [6, 9, 7])):
        samples = synth_batch(rng, lens, 257, 2)
        batch = coll(samples)
        loss, norm = m.compute_loss(model, 0, batch)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
        opt.step()
        fx["step%d_lens" % step] = np.array(lens)
        fx["step%d_loss" % step] = loss.detach().numpy()
        fx["step%d_norm" % step] = norm.detach().numpy()
        fx["step%d_gnorm" % step] = np.array(float(gn))
        fx["step%d_h0" % step], fx["step%d_c0" % step] = rec["h0"].numpy(), rec["c0"].numpy()
        for i, d in enumerate(samples):
            for k, v in d.items():
                fx["step%d_sample%d_%s" % (step, i, k)] = v
    for k, v in param_checksums(model).items():
        fx["wsum_final_" + k] = v
    np.savez_compressed(os.path.join(HERE, "ref_upit_train3.npz"), **fx)

    # ---- case 3: compute_masks in eval mode (running stats) --------------------------------
    torch.manual_seed(99)
    model = m.SepDNN(-1)
    with torch.no_grad():                       # non-trivial running statistics
        model.bn.running_mean.normal_(0.0, 0.05)
        model.bn.running_var.uniform_(0.5, 1.5)
    model.eval()
    rec = capture_hidden(model)
    rng = np.random.default_rng(99)
    lens = [7, 11, 9]
    samples = synth_batch(rng, lens, 257, 2, with_names=True)
    batch = m.Collator("mix")(samples)
    outdir = "/tmp/ref_masks_fixture"
    os.makedirs(outdir, exist_ok=True)
    with torch.no_grad():
        m.compute_masks(model, batch, outdir)
    fx = {"seed": np.array(99), "lens": np.array(lens), "h0": rec["h0"].numpy(), "c0": rec["c0"].numpy(),
          "running_mean": model.bn.running_mean.numpy(), "running_var": model.bn.running_var.numpy(),
          "names": np.array(batch["name"])}
    for i, d in enumerate(samples):
        fx["sample%d_mix" % i] = d["mix"]
        z = np.load(os.path.join(outdir, d["name"]))
        for k in z.files:
            fx["mask_%s_%s" % (d["name"], k)] = z[k]
    np.savez_compressed(os.path.join(HERE, "ref_upit_masks.npz"), **fx)
    print("fixtures written:", out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rsh":
        main_rsh()
    else:
        main()
        main_rsh()
