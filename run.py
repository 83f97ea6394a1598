"""python run.py experiment=conv3d trainer.gpus=1 datamodule.fake_data=true
Entry point with the reference's CLI surface (run.py:16-39): compose configs/config.yaml + groups + overrides,
extras, print_config, train."""
import os

from predict_pv_yield_amd import hydra_lite

os.environ["HYDRA_FULL_ERROR"] = "1"


@hydra_lite.main(config_path="configs/", config_name="config.yaml")
def main(config):
    from predict_pv_yield_amd.training import train
    from predict_pv_yield_amd.utils import extras, print_config

    extras(config)
    if config.get("print_config"):
        print_config(config, resolve=True)
    return train(config)


if __name__ == "__main__":
    main()
