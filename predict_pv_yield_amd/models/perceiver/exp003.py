"""LitModel of the reference's third experiment — host-side mirror of
experiments/003_perceiver_processes_single_sat_image_then_rnn.py:95-253 (BASELINE.json configs[4]).

Every 5-minute satellite image of the stack is one Perceiver example: [B, T, W, H, 12] -> B*T images of W*H positions
(16 384 at 128 x 128 pixels) x 38 channels (12 + 26 Fourier features), cross-attended by 128 latents x 64, depth 2,
layers NOT weight-tied (experiments/003...py:105-114).  The 512 Perceiver outputs go through fc1 (+ PV-system embedding)
.. fc5 to 8 features per time step, which -- with the flattened NWP values (10 x 2 x 2), four datetime features and, for
the encoder, the PV history -- feed a 2-layer GRU encoder over the history and a 2-layer GRU decoder over the forecast
steps; decoder_fc1/2 give one yield per forecast step.  Loss = NMAE, metrics MSE / NMAE (:255-275), Adam 5e-4 (:298).

Same constructor (history_len, forecast_len), same attribute / state_dict names and the same batch keys as the
experiment's dict batches (`sat_data`, `pv_system_row_number`, `nwp`, `hour_of_day_sin|cos`, `day_of_year_sin|cos`,
`pv_yield`).  The experiment trains with `precision=16` under DDP (:40, :288-294): `operand_dtype="bf16"` runs the
attention products on the bf16 matrix cores (f32 accumulation, f32 softmax), "f32" keeps the exact-f32 kernels.
All arithmetic runs in the gfx950 kernels behind include/pv_yield_hip.h; there is no CPU path.
"""
import torch
from torch import nn

from ...lightning import LightningModule
from .perceiver_core import Perceiver

params = dict(
    batch_size=32,
    history_len=6,    #: Number of timesteps of history, not including t0.
    forecast_len=12,  #: Number of timesteps of forecast.
    image_size_pixels=64,
    nwp_channels=("t", "dswrf", "prate", "r", "sde", "si10", "vis", "lcc", "mcc", "hcc"),
    sat_channels=("HRV", "IR_016", "IR_039", "IR_087", "IR_097", "IR_108", "IR_120", "IR_134", "VIS006", "VIS008", "WV_062",
                  "WV_073"),
    precision=16,
)

TOTAL_SEQ_LEN = params["history_len"] + params["forecast_len"] + 1
EMBEDDING_DIM = 16
NWP_SIZE = len(params["nwp_channels"]) * 2 * 2  # channels x width x height
N_DATETIME_FEATURES = 4
PERCEIVER_OUTPUT_SIZE = 512
FC_OUTPUT_SIZE = 8
RNN_HIDDEN_SIZE = 16


class LitModel(LightningModule):
    name = "exp003_perceiver_then_rnn"

    def __init__(self, history_len=params["history_len"], forecast_len=params["forecast_len"], operand_dtype: str = "bf16"):
        super().__init__()
        self.history_len = history_len
        self.forecast_len = forecast_len
        self.total_seq_len = history_len + forecast_len + 1

        self.perceiver = Perceiver(
            input_channels=len(params["sat_channels"]),
            input_axis=2,
            num_freq_bands=6,
            max_freq=10,
            depth=2,
            num_latents=128,
            latent_dim=64,
            num_classes=PERCEIVER_OUTPUT_SIZE,
            operand_dtype=operand_dtype,
        )

        self.fc1 = nn.Linear(in_features=PERCEIVER_OUTPUT_SIZE, out_features=256)
        self.fc2 = nn.Linear(in_features=256 + EMBEDDING_DIM, out_features=128)
        self.fc3 = nn.Linear(in_features=128, out_features=64)
        self.fc4 = nn.Linear(in_features=64, out_features=32)
        self.fc5 = nn.Linear(in_features=32, out_features=FC_OUTPUT_SIZE)
        if EMBEDDING_DIM:
            self.pv_system_id_embedding = nn.Embedding(num_embeddings=940, embedding_dim=EMBEDDING_DIM)
        # plus 1 for history
        self.encoder_rnn = nn.GRU(input_size=FC_OUTPUT_SIZE + N_DATETIME_FEATURES + 1 + NWP_SIZE, hidden_size=RNN_HIDDEN_SIZE,
                                  num_layers=2, batch_first=True)
        self.decoder_rnn = nn.GRU(input_size=FC_OUTPUT_SIZE + N_DATETIME_FEATURES + NWP_SIZE, hidden_size=RNN_HIDDEN_SIZE,
                                  num_layers=2, batch_first=True)
        self.decoder_fc1 = nn.Linear(in_features=RNN_HIDDEN_SIZE, out_features=8)
        self.decoder_fc2 = nn.Linear(in_features=8, out_features=1)

    def forward(self, x):
        from ... import functional as Fn
        from ... import perceiver_functional as PF
        # ******************* Satellite imagery *************************
        # Shape: batch_size, seq_length, width, height, channel
        sat_data = x["sat_data"]
        if not sat_data.is_cuda:
            raise RuntimeError("predict_pv_yield_amd exp003.LitModel runs on the MI355X only: move the module and the batch to "
                               "cuda (there is no CPU fallback)")
        batch_size, seq_len, width, height, n_chans = sat_data.shape
        # Stack timesteps as examples (to make a large batch)
        new_batch_size = batch_size * seq_len
        sat_data = sat_data.reshape(new_batch_size, width, height, n_chans)

        out = self.perceiver(sat_data)
        out = out.reshape(new_batch_size, PERCEIVER_OUTPUT_SIZE)
        out = Fn.linear_f32(out, self.fc1.weight, self.fc1.bias, relu=True)

        # ********************** Embedding of PV system ID ********************
        if EMBEDDING_DIM:
            pv_row = x["pv_system_row_number"].to(dtype=torch.int64).repeat_interleave(self.total_seq_len)
            out = torch.cat((out, Fn.embedding(self.pv_system_id_embedding.weight, pv_row)), dim=1)

        out = Fn.linear_f32(out, self.fc2.weight, self.fc2.bias, relu=True)
        out = Fn.linear_f32(out, self.fc3.weight, self.fc3.bias, relu=True)
        out = Fn.linear_f32(out, self.fc4.weight, self.fc4.bias, relu=True)
        out = Fn.linear_f32(out, self.fc5.weight, self.fc5.bias, relu=True)

        # ******************* PREP DATA FOR RNN *******************************
        out = out.reshape(batch_size, self.total_seq_len, FC_OUTPUT_SIZE)

        # *********************** NWP Data ************************************
        # Shape: batch_size, channel, seq_length, width, height; the RNN expects seq_len to be dim 1
        nwp_data = x["nwp"].float().permute(0, 2, 1, 3, 4)
        batch_size, nwp_seq_len, n_nwp_chans, nwp_width, nwp_height = nwp_data.shape
        nwp_data = nwp_data.reshape(batch_size, nwp_seq_len, n_nwp_chans * nwp_width * nwp_height)

        rnn_input = torch.cat(
            (out, nwp_data, x["hour_of_day_sin"].unsqueeze(-1), x["hour_of_day_cos"].unsqueeze(-1),
             x["day_of_year_sin"].unsqueeze(-1), x["day_of_year_cos"].unsqueeze(-1)), dim=2).float()

        pv_yield_history = x["pv_yield"][:, : self.history_len + 1].unsqueeze(-1).float()
        encoder_input = torch.cat((rnn_input[:, : self.history_len + 1], pv_yield_history), dim=2)

        _, encoder_hidden = PF.gru(encoder_input, self.encoder_rnn)
        decoder_output, _ = PF.gru(rnn_input[:, -self.forecast_len:], self.decoder_rnn, encoder_hidden)
        # decoder_output is shape batch_size, seq_len, rnn_hidden_size
        b, t, h = decoder_output.shape
        decoder_output = Fn.linear_f32(decoder_output.reshape(b * t, h), self.decoder_fc1.weight, self.decoder_fc1.bias, relu=True)
        decoder_output = Fn.linear_f32(decoder_output, self.decoder_fc2.weight, self.decoder_fc2.bias, relu=False)
        return decoder_output.reshape(b, t)

    def _training_or_validation_step(self, batch, is_train_step):
        from ...functional import forecast_losses
        y_hat = self(batch)
        y = batch["pv_yield"][:, -self.forecast_len:].float()
        mse_loss, nmae_loss, _, _ = forecast_losses(y_hat.float(), y)      # one launch; nmae carries the gradient
        tag = "Train" if is_train_step else "Validation"
        self.log_dict({f"MSE/{tag}": mse_loss, f"NMAE/{tag}": nmae_loss}, on_step=is_train_step, on_epoch=True, sync_dist=True)
        return nmae_loss

    def training_step(self, batch, batch_idx):
        return self._training_or_validation_step(batch, is_train_step=True)

    def validation_step(self, batch, batch_idx):
        # the experiment also uploads plots of a few examples to Neptune here (:277-294): outside the hot path
        return self._training_or_validation_step(batch, is_train_step=False)

    def configure_optimizers(self):
        from ...optim import HipAdam
        return HipAdam(self.parameters(), lr=0.0005)


def make_fake_exp003_batch(batch_size: int = 32, image_size_pixels: int = 128, generator=None, history_len=params["history_len"],
                           forecast_len=params["forecast_len"]):
    """Synthetic dict batch with the experiment's keys and shapes (experiments/003...py:139-203)."""
    g = generator
    t = history_len + forecast_len + 1
    phase = torch.rand(batch_size, 1, generator=g) * 6.2831853
    steps = torch.arange(t, dtype=torch.float32)[None] * 0.02
    return {
        "sat_data": torch.randn(batch_size, t, image_size_pixels, image_size_pixels, len(params["sat_channels"]), generator=g),
        "pv_system_row_number": torch.randint(0, 940, (batch_size,), generator=g),
        "nwp": torch.randn(batch_size, len(params["nwp_channels"]), t, 2, 2, generator=g),
        "hour_of_day_sin": torch.sin(phase + steps), "hour_of_day_cos": torch.cos(phase + steps),
        "day_of_year_sin": torch.sin(phase * 0.5 + steps * 0.01), "day_of_year_cos": torch.cos(phase * 0.5 + steps * 0.01),
        "pv_yield": torch.rand(batch_size, t, generator=g),
    }


class FakeExp003Dataset(torch.utils.data.Dataset):
    """Each item is a whole seeded batch (DataLoader(batch_size=None)), like the experiment's own loaders."""

    def __init__(self, batch_size: int = 32, image_size_pixels: int = 128, length: int = 4, seed: int = 1234):
        self.batch_size, self.image_size_pixels, self.length, self.seed = batch_size, image_size_pixels, length, seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        if idx >= self.length:
            raise IndexError(idx)
        return make_fake_exp003_batch(self.batch_size, self.image_size_pixels, torch.Generator().manual_seed(self.seed + idx))
