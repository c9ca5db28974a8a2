"""Imputer entry point, mirror of the reference's ``BaseLine/DiffGANI.py``: same call surface
(``get_Diff_acc_RMSE``), same artefact files, same construction order of the networks so that the
weights initialise identically under the reference's seeds; the N x N attention, the training steps
and the T-step sampler run in the HIP library."""
from __future__ import annotations

import random

import numpy as np
import pandas as pd
import torch

from . import Mean
from ..model import Diffusion, Discriminator_model
from ..utils.get_attr_attn import get_attr_map_Null_kendall
from ..utils.util import Data_convert, Data_convert_device, categorical_to_code, get_M_by_data_m, init_attn_2, sort_corr


def get_Diff_acc_RMSE(nan_data, path, miss_rate, miss_data, enc, data_m, categorical_cols, continuous_cols, value_cat,
                      device, param, label_data, values, ori_data, label_num, args, rng="host", log=None, epoch_hook=None):
    """reference: BaseLine/DiffGANI.py:14-60 -> np.ndarray [N,C] (object dtype with categoricals).
    Mutates ``nan_data`` in place (mean / mode pre-fill) and round-trips the attention-filled table
    through ``Attention_Input_miss_data_<rate>.csv`` like the reference (:23-24)."""
    pathMiss = path + 'miss_data_{}.csv'.format(miss_rate)
    corr_map = get_attr_map_Null_kendall(pathMiss, data_m, categorical_cols, continuous_cols, device=device)
    sort_corr_dict = sort_corr(corr_map)
    attention_data = Mean.fill_data_mean(nan_data, continuous_cols)
    attention_impute_data = pd.DataFrame(attention_data, columns=values)
    if args.UseAttention == 'True':
        attention_impute_data, _ = init_attn_2(corr_map, attention_impute_data, data_m, categorical_cols, enc,
                                               value_cat, device, param['top_k'])
        out_csv = path + 'Attention_Input_miss_data_{}.csv'.format(miss_rate)
        attention_impute_data.to_csv(out_csv, index=None)
        attention_impute_data = pd.read_csv(out_csv)
    d_input_dim = attention_impute_data.shape[1]
    cat_to_code_data, enc = categorical_to_code(attention_impute_data.copy(), value_cat, enc)
    cat_to_code_data.columns = list(range(cat_to_code_data.shape[1]))
    if torch.device(device).type == "cuda":      # (f3) the encode on the device: no host one-hot table
        fields, impute_data_code = Data_convert_device(cat_to_code_data, param['model_name'], continuous_cols, device)
        feed_data = impute_data_code
    else:
        fields, feed_data = Data_convert(cat_to_code_data, param['model_name'], continuous_cols)
        impute_data_code = torch.tensor(feed_data.values, dtype=torch.float).to(device)
    if args.UseFD == "True":
        # reference :30-39: dependencies are discovered on the table with missing cells, the classifiers are trained
        # on the observed part of the encoded table (M * code).  Their loss is a no_grad constant in the generator
        # loss: numerically inert for the imputation, but it is what the flag computes.
        from ..model.FD_model import get_FD_model_Tree
        M_tensor = get_M_by_data_m(data_m, fields, device)
        FD_model_list = get_FD_model_Tree(miss_data, data_m, categorical_cols, impute_data_code * M_tensor, fields,
                                          device, sort_corr_dict)
    else:
        FD_model_list = []
    input_dim = feed_data.shape[1]
    latent_dim = input_dim // 2
    random.choice([500, 400, 300, 200, 100])              # the reference draws an unused width here (:45)
    torch.manual_seed(3407)
    num_steps = param['T']
    Diffusion.consume_dead_module_rng(input_dim, num_steps + 1)
    Discriminator = Discriminator_model.D(input_dim, latent_dim, d_input_dim)
    Discriminator_denoise_x = Discriminator_model.Discriminator_noise_x(input_dim * 2, latent_dim, d_input_dim,
                                                                        num_steps + 1)
    Generator_x0 = Diffusion.Generator_x0(input_dim, input_dim * 2, input_dim, num_steps + 1, fields)
    return Diffusion.train_diffusion_discriminator(Discriminator, Generator_x0, Discriminator_denoise_x, FD_model_list,
                                                   num_steps, param["epochs"], param["lr"], param['batch_size'],
                                                   param["loss_weight"], data_m, impute_data_code, label_data, fields,
                                                   value_cat, values, attention_impute_data, enc, ori_data,
                                                   continuous_cols, label_num, device, args.UseLearner, rng=rng,
                                                   log=log, epoch_hook=epoch_hook)
