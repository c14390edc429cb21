"""FlowMatchDiscreteScheduler with the reference's call surface
(hyvideo/diffusion/schedulers/scheduling_flow_match_discrete.py:48-257): shifted-sigma schedule on the host
(51 floats) and the fp32 Euler update as one HBM-bound kernel (hv_euler_step_f32)."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch

from ... import ops


class FlowMatchDiscreteScheduler:
    _compatibles = []
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, reverse: bool = True, solver: str = "euler",
                 n_tokens: Optional[int] = None):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, reverse=reverse,
                                      solver=solver, n_tokens=n_tokens)
        sigmas = torch.linspace(1, 0, num_train_timesteps + 1)
        if not reverse:
            sigmas = sigmas.flip(0)
        self.sigmas = sigmas
        self.timesteps = (sigmas[:-1] * num_train_timesteps).to(dtype=torch.float32)
        self._step_index = None
        self._begin_index = None
        self.supported_solver = ["euler"]
        if solver not in self.supported_solver:
            raise ValueError(f"Solver {solver} not supported. Supported solvers: {self.supported_solver}")

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def _sigma_to_t(self, sigma):
        return sigma * self.config.num_train_timesteps

    def sd3_time_shift(self, t: torch.Tensor):
        return (self.config.shift * t) / (1 + (self.config.shift - 1) * t)

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device] = None, n_tokens: int = None):
        self.num_inference_steps = num_inference_steps
        sigmas = self.sd3_time_shift(torch.linspace(1, 0, num_inference_steps + 1))
        if not self.config.reverse:
            sigmas = 1 - sigmas
        self.sigmas = sigmas                       # host copy: dt is a launch argument, never a device read
        self._host_timesteps = (sigmas[:-1] * self.config.num_train_timesteps).to(torch.float32)
        self.timesteps = self._host_timesteps.to(device=device)
        self._step_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        ts = self._host_timesteps if schedule_timesteps is None else schedule_timesteps.to("cpu")
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.to("cpu")
        indices = (ts == timestep).nonzero()
        pos = 1 if len(indices) > 1 else 0
        return indices[pos].item()

    def _init_step_index(self, timestep):
        self._step_index = self.index_for_timestep(timestep) if self.begin_index is None else self._begin_index

    def scale_model_input(self, sample: torch.Tensor, timestep: Optional[int] = None) -> torch.Tensor:
        return sample

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True) -> Tuple:
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to"
                             " `EulerDiscreteScheduler.step()` is not supported. Make sure to pass"
                             " one of the `scheduler.timesteps` as a timestep.")
        if self.step_index is None:
            self._init_step_index(timestep)
        dt = float(self.sigmas[self.step_index + 1] - self.sigmas[self.step_index])
        # upcast (a no-op from step 2 on: latents are fp32 after the first step); out of place like the reference (:236-242).
        # `prev` must be a fresh CONTIGUOUS buffer before the in-place kernel: .to()/.clone() keep a permuted sample's strides,
        # and a later .contiguous() would update a temporary and silently drop the step.
        prev = sample.to(torch.float32).contiguous()
        if prev.data_ptr() == sample.data_ptr():
            prev = prev.clone()
        if model_output.dtype == torch.bfloat16:
            ops.euler_step_(prev, model_output.contiguous(), dt)
        else:
            # the reference upcasts the model output to fp32 (:239); a non-bf16 output keeps that precision here
            ops.euler_step_f32_(prev, model_output.to(torch.float32).contiguous(), dt)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return SimpleNamespace(prev_sample=prev)

    def __len__(self):
        return self.config.num_train_timesteps
