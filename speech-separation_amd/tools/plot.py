#!/usr/bin/env python3
"""Diagnostic plots used by the arch modules and the train driver (same entry points as the
reference's tools/plot.py: plot_spec(array, path), plot_loss(train, [cv,] path)).  Plain
matplotlib (Agg); off the hot path."""
import os

import numpy as np
import matplotlib
matplotlib.use('Agg')
import matplotlib.pyplot as plt  # noqa: E402


def plot_spec(array, path):
  fig, ax = plt.subplots()
  im = ax.imshow(np.flipud(np.asarray(array).T), aspect='auto')
  ax.set_xticks([])
  ax.set_yticks([])
  ax.set_xlabel('time')
  ax.set_ylabel('frequency')
  ax.set_title(os.path.basename(path).split('.')[0].replace('_', ' '))
  fig.colorbar(im, ax=ax, aspect=40, pad=0.025)
  fig.savefig(path, dpi=200, bbox_inches='tight')
  plt.close(fig)


def plot_loss(*args):
  """plot_loss(train_losses, path) or plot_loss(train_losses, cv_losses, path); each losses
  argument is [epochs, values]."""
  path = args[-1]
  fig, ax = plt.subplots()
  labels = ['train', 'cv']
  for i, series in enumerate(args[:-1]):
    if series and len(series[0]):
      ax.plot(series[0], series[1], label=labels[min(i, 1)])
  ax.set_xlabel('epoch')
  ax.set_ylabel('loss')
  ax.legend(frameon=False)
  fig.savefig(path, dpi=200, bbox_inches='tight')
  plt.close(fig)
