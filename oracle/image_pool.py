"""TEST INFRASTRUCTURE (checker only; the product never imports this): CPU restatement of the reference's history buffer of generated
images, util/image_pool.py:10-61 (pix2pixHD's fake_pool, models/pix2pixHD_model.py:334, 582).  Pinned by tests/golden/image_pool.npz
(the ids the reference's ImagePool returns under a seeded `random`)."""
import random

import torch


def new_pool(pool_size):
    return {"size": int(pool_size), "slots": []}


def pool_query(pool, images):
    """util/image_pool.py:29-61.  Image by image: a pool that is not full stores and returns the image; a full one draws
    random.uniform(0, 1) and, above 0.5, hands out the content of slot random.randint(0, size - 1) and stores the image there."""
    if pool["size"] == 0:
        return images
    back = []
    for img in images:
        img = img.detach()[None]
        if len(pool["slots"]) < pool["size"]:
            pool["slots"].append(img)
            back.append(img)
            continue
        if random.uniform(0, 1) > 0.5:
            k = random.randint(0, pool["size"] - 1)
            back.append(pool["slots"][k].clone())
            pool["slots"][k] = img
        else:
            back.append(img)
    return torch.cat(back, 0)
