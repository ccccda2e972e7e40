#!/usr/bin/env python3
"""Wall-clock of every stage of the recipe THROUGH the drop-in drivers, on a synthetic WSJ0-2mix-shaped corpus.

    python speech-separation_amd/tools/stage_walls.py [--utts 2000] [--work /tmp/sk_walls] [--hidden 896 --layers 3]

Stages (the reference's run_train.sh / run_eval.sh order): steps/extract_feats.py (train + test features),
steps/train_qsub.py -- two epochs, the second one is reported -- in three loader modes:
    reference loop   npz features, --num-workers 1 --prefetch 0   (steps/train_qsub.py:80-84,113-122 of the reference)
    npz, staged      npz features, default workers, batches staged on the GPU two ahead
    wav, staged      --wav-input: int16 PCM through the loader, STFT on the GPU's copy stream
then steps/eval_qsub.py (masks) and steps/reconstruct_sources.py (mask-apply + iSTFT + wav files).
Every driver prints its own frames/s to stderr; this script runs them as child processes and collects those lines
next to the wall time of the whole process (interpreter start, imports and model build included).
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, PKG)


def run(tag, cmd, env, out):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    lines = [l for l in r.stderr.splitlines() if re.search(r"frames/s|skipped|timed out|prefetch:", l)]
    out.append("%-34s %8.2f s   rc %d" % (tag, dt, r.returncode))
    for l in lines:
        out.append("      " + l.strip())
    if r.returncode != 0:
        out.append("      STDERR TAIL: " + r.stderr[-1500:].replace("\n", "\n      "))
    print(out[-1 - len(lines)], flush=True)
    for l in lines:
        print("      " + l.strip(), flush=True)
    return r.returncode == 0, lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2000)
    ap.add_argument("--eval-utts", type=int, default=400)
    ap.add_argument("--work", default="/tmp/sk_walls")
    ap.add_argument("--hidden", type=int, default=896)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default="")
    ap.add_argument("--modes", default="ref,npz,wav")
    ap.add_argument("--conf", default="", help="extra conf lines, comma separated (e.g. dtype=bf16)")
    ap.add_argument("--epochs", type=int, default=2, help="training epochs per mode (the last one is the reported rate; > 2: a soak run)")
    ap.add_argument("--train-args", default="", help="extra arguments for steps/train_qsub.py, space separated (e.g. '--num-workers 6')")
    ap.add_argument("--skip-eval", action="store_true", help="training stages only")
    ap.add_argument("--keep", action="store_true", help="leave the work directory (tools/loader_probe.py reads its data dirs)")
    a = ap.parse_args()
    from sepkern import synth
    work = os.path.abspath(a.work)
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    out = ["stage walls through the drop-in drivers: %d training / %d evaluation utterances of U(3 s, 8 s) at 8 kHz, "
           "uPIT %dx%d, batch %d" % (a.utts, a.eval_utts, a.layers, a.hidden, a.batch)]
    t0 = time.perf_counter()
    tr_wav, te_wav = os.path.join(work, "wav_tr"), os.path.join(work, "wav_tt")
    ids = synth.write_wav_tree(tr_wav, a.utts, num_spk=2, seed=1)
    synth.write_data_dir(os.path.join(work, "data", "tr"), tr_wav, ids)
    ids_t = synth.write_wav_tree(te_wav, a.eval_utts, num_spk=2, seed=2)
    synth.write_data_dir(os.path.join(work, "data", "tt"), te_wav, ids_t)
    out.append("%-34s %8.2f s   (not a stage of the recipe)" % ("synthetic corpus written", time.perf_counter() - t0))
    print(out[-1], flush=True)
    env = dict(os.environ, SEPKERN_HOME=PKG, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    py, steps = sys.executable, os.path.join(PKG, "steps")
    d_tr, d_tt = os.path.join(work, "data", "tr"), os.path.join(work, "data", "tt")
    run("extract_feats.py train", [py, os.path.join(steps, "extract_feats.py"), d_tr, "train", os.path.join(work, "feats", "tr_train")], env, out)
    run("extract_feats.py test", [py, os.path.join(steps, "extract_feats.py"), d_tt, "test", os.path.join(work, "feats", "tt_test")], env, out)
    conf = os.path.join(work, "conf")
    with open(conf, "w") as f:
        f.write("hidden_dim=%d\nnum_layers=%d\nnum_spk=2\n" % (a.hidden, a.layers))
        for line in a.conf.split(","):
            if line:
                f.write(line + "\n")
    base = [py, os.path.join(steps, "train_qsub.py"), "uPIT", "0", d_tr]
    common = ["--model-config", conf, "--batch-size", str(a.batch), "--num-epochs", str(a.epochs), "--seed", "3"] + a.train_args.split()
    modes = {"ref": ("train_qsub.py reference loop", ["--num-workers", "1", "--prefetch", "0"]),
             "npz": ("train_qsub.py npz, staged", []),
             "wav": ("train_qsub.py wav-input, staged", ["--wav-input"])}
    rates = {}
    for m in a.modes.split(","):
        tag, extra = modes[m]
        exp = os.path.join(work, "exp_" + m)
        ok, lines = run(tag, base + [exp] + common + extra, env, out)
        for l in lines:
            g = re.search(r"epoch %d: .* = (\d+) frames/s" % a.epochs, l)
            if g:
                rates[m] = int(g.group(1))
    exp = os.path.join(work, "exp_" + a.modes.split(",")[-1])
    masks = os.path.join(exp, "output", "masks")
    if not a.skip_eval:
        run("eval_qsub.py (masks)", [py, os.path.join(steps, "eval_qsub.py"), os.path.join(PKG, "archs", "uPIT.py"), "0",
                                     os.path.join(exp, "final.mdl"), d_tt, masks, "--model-config", conf, "--batch-size", str(a.batch)], env, out)
        run("reconstruct_sources.py", [py, os.path.join(steps, "reconstruct_sources.py"), d_tt, os.path.join(exp, "output")], env, out)
    out.append("last-epoch training rates (frames/s): " + ", ".join("%s %d" % kv for kv in rates.items()))
    print(out[-1], flush=True)
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(out) + "\n")
    if not a.keep:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
