"""experiment/metrics.py:5-44 — end accuracy / forgetting / ACC / BWT+ / FWT with 95% t-confidence intervals,
applied to the accuracy arrays gathered from all ranks (one independent run per GPU)."""
import numpy as np
from scipy.stats import sem
import scipy.stats as stats


def compute_performance(end_task_acc_arr):
    """end_task_acc_arr: [n_run, n_tasks, n_tasks] (accuracy on task j after training task i)."""
    n_run, n_tasks = end_task_acc_arr.shape[:2]
    t_coef = stats.t.ppf((1 + 0.95) / 2, n_run - 1)

    end_acc = end_task_acc_arr[:, -1, :]
    per_run = np.mean(end_acc, axis=1)
    avg_end_acc = (np.mean(per_run), t_coef * sem(per_run))

    forgets = np.max(end_task_acc_arr, axis=1) - end_acc
    fgt = np.mean(forgets, axis=1)
    avg_end_fgt = (np.mean(fgt), t_coef * sem(fgt))

    acc_per_run = np.mean(np.sum(np.tril(end_task_acc_arr), axis=2) / (np.arange(n_tasks) + 1), axis=1)
    avg_acc = (np.mean(acc_per_run), t_coef * sem(acc_per_run))

    denom = n_tasks * (n_tasks - 1) / 2
    bwt = (np.sum(np.tril(end_task_acc_arr, -1), axis=(1, 2)) -
           np.sum(np.diagonal(end_task_acc_arr, axis1=1, axis2=2) * (np.arange(n_tasks, 0, -1) - 1), axis=1)) / denom
    bwtp = np.maximum(bwt, 0)
    avg_bwtp = (np.mean(bwtp), t_coef * sem(bwtp))

    fwt = np.sum(np.triu(end_task_acc_arr, 1), axis=(1, 2)) / denom
    avg_fwt = (np.mean(fwt), t_coef * sem(fwt))
    return avg_end_acc, avg_end_fgt, avg_acc, avg_bwtp, avg_fwt
