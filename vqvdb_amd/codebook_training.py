"""Codebook (EMA) training on the HIP backend — host side of SURVEY.md §8 f-2, stage 1.

Mirrors what the reference's training loop does for the quantizer (python/training.py:47-258 drives
VectorQuantizerEMA.forward in training mode, python/VQVAE_v2.py:107-156, and check_and_reset_dead_codes,
:382-417), with the encoder and decoder weights frozen.  The device work is in libvqvdb_hip.so
(vqhip_train_*); this module is the data-parallel plumbing around it:

    per step and rank:  stats = encoder -> latent -> assign -> {encodings_sum, dw, |z-e|^2, rows}   (HIP kernels)
                        all_reduce(stats, SUM)          # RCCL over xGMI; the only collective on the path
                        EMA update of cluster_size / embed_avg / embedding from the global stats    (HIP kernel)

Every rank applies the identical update, so codebooks stay replicated without a broadcast.  torch is used for device
memory, streams and torch.distributed only.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

K, D = 256, 128
STATS_FLOATS = K + K * D + K + 1
_DW, _SQ, _ROWS = K, K + K * D, K + K * D + K


def metrics_from_stats(stats: np.ndarray, commitment_cost: float = 0.25) -> dict:
    """vq_loss = commitment_cost * mse(z, quantized) (VQVAE_v2.py:146) and perplexity (:153-154) from the
    (all-reduced) statistics buffer."""
    stats = np.asarray(stats, dtype=np.float64)
    rows = stats[_ROWS]
    counts = stats[:K]
    p = counts / rows
    return {"rows": int(rows), "vq_loss": float(commitment_cost * stats[_SQ:_SQ + K].sum() / (rows * D)),
            "perplexity": float(np.exp(-(p * np.log(p + 1e-10)).sum())), "codes_used": int((counts > 0).sum())}


def allreduce_stats(stats: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank statistics in place (no-op without an initialised process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def dead_code_reset(state: dict, flat_z: torch.Tensor, threshold: float = 1.0, generator: Optional[torch.Generator] = None,
                    group=None, src: int = 0) -> int:
    """check_and_reset_dead_codes (VQVAE_v2.py:382-417): codes with cluster_size < threshold are re-sampled from the given
    encoder outputs (uniform row indices), their embed_avg set to the same rows and cluster_size to 1.  `state` holds torch
    tensors embedding [256,128], cluster_size [256], embed_avg [256,128] and is modified in place.  With a process group,
    rank `src` draws the samples and the three buffers are broadcast (the draw uses that rank's RNG)."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    n_dead = 0
    if not distributed or dist.get_rank(group) == src:
        dead = torch.where(state["cluster_size"] < threshold)[0]
        n_dead = int(dead.numel())
        if n_dead and flat_z.shape[0]:
            pick = torch.randint(0, flat_z.shape[0], (n_dead,), device=flat_z.device, generator=generator)
            new = flat_z[pick].to(state["embedding"].device, torch.float32)
            state["embedding"][dead] = new
            state["embed_avg"][dead] = new
            state["cluster_size"][dead] = 1.0
    if distributed:
        n = torch.tensor([n_dead], device=state["embedding"].device)
        dist.broadcast(n, src=src, group=group)
        n_dead = int(n.item())
        if n_dead:
            for k in ("embedding", "cluster_size", "embed_avg"):
                dist.broadcast(state[k], src=src, group=group)
    return n_dead


class CodebookTrainer:
    """Drives vqhip_train_* for one rank.  `codec` is a vqvdb_amd.codec.HipCodec on this rank's device."""

    def __init__(self, codec, commitment_cost: float = 0.25, decay: float = 0.95, eps: float = 1e-4, group=None,
                 cluster_size: Optional[np.ndarray] = None, embed_avg: Optional[np.ndarray] = None, device: str = "cuda"):
        self.codec, self.group = codec, group
        self.commitment_cost, self.decay, self.eps = commitment_cost, decay, eps
        self.device = torch.device(device)
        codec.train_begin(cluster_size, embed_avg)
        self.stats = torch.zeros(STATS_FLOATS, dtype=torch.float32, device=self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.latent = None   # encoder outputs of the last step that asked for them (dead-code reset input)

    def step(self, leaves: torch.Tensor, keep_latent: bool = False, want_metrics: bool = True) -> Optional[dict]:
        """One EMA step on this rank's batch (float32 [n,512] or [n,1,8,8,8], resident on the device)."""
        leaves = leaves.contiguous()
        if leaves.dtype != torch.float32 or leaves.numel() % 512:
            raise ValueError("leaves must be float32 with 512 values per leaf")
        n = leaves.numel() // 512
        # The library treats a NULL stream as "use the codec's own stream", so torch's default (null) stream cannot be
        # handed over: run the step on a side stream ordered after the producer of `leaves` and before later consumers.
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        out = None
        with torch.cuda.stream(self.stream):
            zptr = 0
            if keep_latent:
                if self.latent is None or self.latent.shape[0] != n * 64:
                    self.latent = torch.empty((n * 64, D), dtype=torch.float32, device=self.device)
                zptr = self.latent.data_ptr()
            h = self.stream.cuda_stream
            self.codec.train_vq_stats_device(leaves.data_ptr(), n, self.stats.data_ptr(), latent_ptr=zptr, stream=h)
            allreduce_stats(self.stats, self.group)
            self.codec.train_vq_update_device(self.stats.data_ptr(), self.decay, self.eps, stream=h)
            if want_metrics:
                out = metrics_from_stats(self.stats.cpu().numpy(), self.commitment_cost)
        leaves.record_stream(self.stream)
        cur.wait_stream(self.stream)
        return out

    def evaluate(self, leaves: torch.Tensor, mse_weight: float = 0.8, l1_weight: float = 0.2) -> dict:
        """Validation forward on this rank's batch (training.py:183-199): reconstruction MSE / L1 (and the reference's
        0.8 / 0.2 mix, :151-155), vq_loss and perplexity over the GLOBAL batch; nothing is updated."""
        leaves = leaves.contiguous()
        if leaves.dtype != torch.float32 or leaves.numel() % 512:
            raise ValueError("leaves must be float32 with 512 values per leaf")
        n = leaves.numel() // 512
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            buf = torch.zeros(STATS_FLOATS + 3, dtype=torch.float32, device=self.device)
            self.codec.train_eval_device(leaves.data_ptr(), n, buf.data_ptr(), buf[STATS_FLOATS:].data_ptr(), stream=self.stream.cuda_stream)
            allreduce_stats(buf, self.group)
            host = buf.cpu().numpy().astype(np.float64)
        leaves.record_stream(self.stream)
        cur.wait_stream(self.stream)
        out = metrics_from_stats(host[:STATS_FLOATS], self.commitment_cost)
        sq, ab, elems = host[STATS_FLOATS:]
        out.update(recon_mse=float(sq / elems), recon_l1=float(ab / elems))
        out["recon_error"] = mse_weight * out["recon_mse"] + l1_weight * out["recon_l1"]
        return out

    def reset_dead_codes(self, flat_z: Optional[torch.Tensor] = None, threshold: float = 1.0, generator=None) -> int:
        flat_z = self.latent if flat_z is None else flat_z
        if flat_z is None:
            raise ValueError("no encoder outputs kept: call step(..., keep_latent=True) first or pass flat_z")
        st = {k: torch.from_numpy(v).to(self.device) for k, v in self.codec.train_get_state().items()}
        n = dead_code_reset(st, flat_z, threshold, generator, self.group)
        if n:
            self.codec.train_set_state(**{k: v.cpu().numpy() for k, v in st.items()})
        return n

    def state_dict(self) -> dict:
        """quantizer.* buffers in the reference's state_dict naming (VQVAE_v2.py:103-105)."""
        return {f"quantizer.{k}": v for k, v in self.codec.train_get_state().items()}

    def load_state_dict(self, sd: dict):
        """Resume from quantizer.* buffers saved by state_dict() or by the reference's checkpoints (training.py:216-233)."""
        self.codec.train_set_state(embedding=sd["quantizer.embedding"], cluster_size=sd["quantizer.cluster_size"],
                                   embed_avg=sd["quantizer.embed_avg"])

    def finish(self):
        """Refresh the inference tables (folded search, decoder stem table) from the trained codebook."""
        self.codec.train_commit()
