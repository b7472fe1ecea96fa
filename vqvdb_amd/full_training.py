"""Full VQ-VAE training step on the HIP backend — host side of SURVEY.md §8 f-2, stage 2 (python/training.py:47-258).

Per step and rank:  forward + backward of the rank's batch -> flat gradient vector G and the auxiliary sums (HIP kernels)
                    all_reduce(G, SUM), all_reduce(aux, SUM)          # RCCL over xGMI: 3.98 MB + 133 KB
                    AdamW on the 995 905 parameters, EMA on the codebook, device tables rebuilt (HIP kernels)
The loss is the reference's 0.8 mse + 0.2 l1 + vq_loss in fp32 (no autocast); gradients are those of the mean over the GLOBAL
batch, so every rank applies the identical update and the replicas never diverge.  torch is used for device memory, streams,
torch.distributed and the cosine learning-rate schedule only.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from vqvdb_amd import synth
from vqvdb_amd.codebook_training import STATS_FLOATS, allreduce_stats, dead_code_reset, metrics_from_stats

AUX_FLOATS = STATS_FLOATS + 3
TRAINABLE = [(name, shape) for name, shape, _ in synth.TENSORS if not name.startswith("quantizer.")]


def flat_to_dict(flat: np.ndarray) -> dict:
    """Flat parameter vector (the reference's parameter order) -> {state_dict name: array}."""
    out, off = {}, 0
    for name, shape in TRAINABLE:
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape).copy()
        off += n
    if off != flat.size:
        raise ValueError("flat parameter vector has the wrong length")
    return out


def dict_to_flat(params: dict) -> np.ndarray:
    return np.concatenate([np.asarray(params[name], dtype=np.float32).reshape(-1) for name, _ in TRAINABLE])


def cosine_lr(base_lr: float, step: int, t_max: int, eta_min: float = 0.0) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingLR (training.py:107-108) in closed form; `step` counts optimizer steps done."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * step / t_max)) / 2.0


class FullTrainer:
    """Drives vqhip_fulltrain_* for one rank.  `codec` is a vqvdb_amd.codec.HipCodec on this rank's device."""

    def __init__(self, codec, lr: float = 1e-4, betas=(0.9, 0.999), adam_eps: float = 1e-8, weight_decay: float = 1e-4,
                 commitment_cost: float = 0.25, ema_decay: float = 0.95, ema_eps: float = 1e-4, t_max: Optional[int] = None, group=None,
                 device: str = "cuda", overlap: bool = True):
        self.codec, self.group, self.device = codec, group, torch.device(device)
        self.lr, self.betas, self.adam_eps, self.weight_decay = lr, betas, adam_eps, weight_decay
        self.commitment_cost, self.ema_decay, self.ema_eps, self.t_max = commitment_cost, ema_decay, ema_eps, t_max
        if abs(commitment_cost - 0.25) > 1e-12:
            raise ValueError("the kernels fix commitment_cost = 0.25 (training.py:55)")
        codec.fulltrain_begin()
        self.grads = torch.zeros(codec.fulltrain_param_count(), dtype=torch.float32, device=self.device)
        self.aux = torch.zeros(AUX_FLOATS, dtype=torch.float32, device=self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.comm_stream = torch.cuda.Stream(device=self.device)   # all-reduce of the decoder's gradients, overlapped with the encoder backward
        self.overlap = overlap
        self.steps_done = 0

    def _world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def step(self, leaves: torch.Tensor, want_metrics: bool = True) -> Optional[dict]:
        """One optimizer step on this rank's batch (float32, 512 values per leaf, resident on the device; every rank passes the
        same number of leaves)."""
        leaves = leaves.contiguous()
        if leaves.dtype != torch.float32 or leaves.numel() % 512:
            raise ValueError("leaves must be float32 with 512 values per leaf")
        n = leaves.numel() // 512
        world = self._world()
        lr = self.lr if self.t_max is None else cosine_lr(self.lr, self.steps_done, self.t_max)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        out = None
        with torch.cuda.stream(self.stream):
            h = self.stream.cuda_stream
            if world > 1 and self.overlap:
                # the decoder's gradients (69 % of the vector) are final once the decoder half of the backward pass is enqueued: their
                # all-reduce runs on its own stream while the encoder half computes; encoder slice + statistics follow on this stream
                dec = self.codec.fulltrain_decoder_offset()
                pending = []

                def decoder_done():
                    ev = torch.cuda.Event()
                    ready = self.codec.fulltrain_ready_stream()   # the codec's second stream carries the weight gradients
                    ev.record(torch.cuda.ExternalStream(ready, device=self.device) if ready else self.stream)
                    self.comm_stream.wait_event(ev)
                    with torch.cuda.stream(self.comm_stream):
                        pending.append(dist.all_reduce(self.grads[dec:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.codec.fulltrain_fwdbwd_overlap_device(leaves.data_ptr(), n, n * world, self.grads.data_ptr(), self.aux.data_ptr(), h, decoder_done)
                dist.all_reduce(self.grads[:dec], op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(self.aux, op=dist.ReduceOp.SUM, group=self.group)
                for w in pending:
                    w.wait()                                  # this stream waits for the decoder slice
                self.stream.wait_stream(self.comm_stream)
            else:
                self.codec.fulltrain_fwdbwd_device(leaves.data_ptr(), n, n * world, self.grads.data_ptr(), self.aux.data_ptr(), stream=h)
                if world > 1:
                    dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=self.group)
                    dist.all_reduce(self.aux, op=dist.ReduceOp.SUM, group=self.group)
            self.codec.fulltrain_apply_device(self.grads.data_ptr(), self.aux.data_ptr(), lr, self.steps_done + 1, self.betas, self.adam_eps,
                                              self.weight_decay, self.ema_decay, self.ema_eps, stream=h)
            if want_metrics:
                aux = self.aux.cpu().numpy().astype(np.float64)
                out = metrics_from_stats(aux[:STATS_FLOATS], self.commitment_cost)
                sq, ab, vox = aux[STATS_FLOATS:]
                out.update(recon_mse=float(sq / vox), recon_l1=float(ab / vox), lr=lr)
                out["recon_error"] = 0.8 * out["recon_mse"] + 0.2 * out["recon_l1"]
                out["loss"] = out["recon_error"] + out["vq_loss"]
        leaves.record_stream(self.stream)
        cur.wait_stream(self.stream)
        self.steps_done += 1
        return out

    def evaluate(self, leaves: torch.Tensor, mse_weight: float = 0.8, l1_weight: float = 0.2) -> dict:
        """Validation forward (training.py:183-199) with the current weights through the inference kernels: reconstruction MSE / L1,
        vq_loss and perplexity over the GLOBAL batch; nothing is updated."""
        leaves = leaves.contiguous()
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

    def reset_dead_codes(self, leaves: torch.Tensor, threshold: float = 1.0, generator=None) -> int:
        """check_and_reset_dead_codes (VQVAE_v2.py:382-417) from the encoder outputs of `leaves` under the current weights."""
        leaves = leaves.contiguous()
        n = leaves.numel() // 512
        z = torch.empty((n * 64, 128), dtype=torch.float32, device=self.device)
        scratch = torch.zeros(STATS_FLOATS, dtype=torch.float32, device=self.device)
        with torch.cuda.stream(self.stream):
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            self.codec.train_vq_stats_device(leaves.data_ptr(), n, scratch.data_ptr(), latent_ptr=z.data_ptr(), stream=self.stream.cuda_stream)
        self.stream.synchronize()
        st = {k: torch.from_numpy(v).to(self.device) for k, v in self.codec.train_get_state().items()}
        k = dead_code_reset(st, z, threshold, generator, self.group)
        if k:
            self.codec.train_set_state(**{kk: v.cpu().numpy() for kk, v in st.items()})
        return k

    def state_dict(self) -> dict:
        """Model state in the reference's state_dict naming (parameters + quantizer buffers)."""
        sd = flat_to_dict(self.codec.fulltrain_get_params())
        sd.update({f"quantizer.{k}": v for k, v in self.codec.train_get_state().items()})
        return sd

    def load_state_dict(self, sd: dict):
        self.codec.fulltrain_set_params(dict_to_flat(sd))
        self.codec.train_set_state(embedding=sd["quantizer.embedding"], cluster_size=sd.get("quantizer.cluster_size"),
                                   embed_avg=sd.get("quantizer.embed_avg"))

    def checkpoint(self) -> dict:
        """Everything a resumed run needs, like the reference's checkpoint (training.py:216-226: model, optimizer and scheduler
        state): state_dict() + AdamW moments + the step count (bias correction and position on the cosine schedule)."""
        ck = self.state_dict()
        m, v = self.codec.fulltrain_get_opt_state()
        ck.update({"optimizer.exp_avg": m, "optimizer.exp_avg_sq": v, "optimizer.steps_done": np.int64(self.steps_done)})
        return ck

    def load_checkpoint(self, ck: dict):
        self.load_state_dict({k: v for k, v in ck.items() if not k.startswith("optimizer.")})
        if "optimizer.exp_avg" in ck:
            self.codec.fulltrain_set_opt_state(ck["optimizer.exp_avg"], ck["optimizer.exp_avg_sq"])
            self.steps_done = int(ck["optimizer.steps_done"])

    def finish(self):
        """Rebuild what the inference path folds on the host (decoder tail, projection inside the VQ search, stem table)."""
        self.codec.train_commit()
