"""Model dimensions read at construction time.

Mirrors the keys of the reference's module-level dict that the hot path reads
(reference configs/model_config.py:3-33).  As in the reference, the modules in
``modules.py`` read this dict when they are *constructed*, so a caller may
re-dimension the model by editing ``model_config`` (or by passing an explicit
``Dims``) before building a ``UserModel``.
"""
from __future__ import annotations

from dataclasses import dataclass

model_config = {
    # reference configs/model_config.py:4  (ids 0..2999, 0 = padding)
    "category_label_num": 3000,
    # reference configs/model_config.py:5  (only its length, 3, reaches the hot path)
    "sentiment_label_dict": {"Negative": 0, "Neutral": 1, "Positive": 2},
    # reference configs/model_config.py:6-22 (only its length, 16, reaches the hot path)
    "article_type_dict": {
        "article_default": 0, "article_webtv": 1, "article_page_nine_girl": 2,
        "article_questions_and_answers": 3, "article_feature": 4, "article_opinionen": 5,
        "article_native": 6, "article_scribblelive": 7, "article_fullscreen_gallery": 8,
        "article_editorial_production": 9, "article_standard_feature": 10,
        "article_native_feature": 11, "article_accordion": 12,
        "article_video_standalone": 13, "article_image_gallery": 14, "article_timeline": 15,
    },
    "pca_vector": 64,            # reference configs/model_config.py:29
    "subcategory_max_num": 5,    # reference configs/model_config.py:30
    "history_max_num": 200,      # reference configs/model_config.py:31
    "inview_max_num": 15,        # reference configs/model_config.py:32
}

# widths of the four time tables (reference models/user_invariant_interest_model.py:34-45)
TIME_TABLE_ROWS = (100, 13, 32, 24)


@dataclass(frozen=True)
class Dims:
    """All sizes the hot path needs, resolved once.

    P   = pca_vector (text+image vector width)
    E   = embed_setting [e0,e1,e2,e3]; D_l = sum(E) is the label-vector width
    """
    pca_vector: int = 64
    embed_setting: tuple = (32, 16, 8, 8)
    category_label_num: int = 3000
    n_sentiment: int = 3
    n_type: int = 16
    n_subcat: int = 5
    instant_dim: int = 8

    @property
    def label_dim(self) -> int:
        return int(sum(self.embed_setting))

    @property
    def history_cols(self) -> int:          # time4 | text_img P | cat1 | sub | sentiment | type1 | read1 | scroll1
        return 4 + self.pca_vector + 1 + self.n_subcat + self.n_sentiment + 1 + 1 + 1

    @property
    def target_cols(self) -> int:           # same minus read_time, scroll
        return self.history_cols - 2

    @property
    def head_dim(self) -> int:              # BatchNorm1d width (reference models/user_model.py:18)
        return (self.label_dim + self.pca_vector) * 2 + self.instant_dim

    @staticmethod
    def from_config(embed_setting=None, cfg=None) -> "Dims":
        cfg = model_config if cfg is None else cfg
        es = (32, 16, 8, 8) if embed_setting is None else tuple(int(e) for e in embed_setting)
        return Dims(
            pca_vector=int(cfg["pca_vector"]),
            embed_setting=es,
            category_label_num=int(cfg["category_label_num"]),
            n_sentiment=len(cfg["sentiment_label_dict"]),
            n_type=len(cfg["article_type_dict"]),
            n_subcat=int(cfg["subcategory_max_num"]),
        )

    @staticmethod
    def for_emb(emb: int, category_label_num: int = 3000) -> "Dims":
        """BASELINE.json shapes: P = D_l = emb, E = [emb/2, emb/4, emb/8, emb/8]."""
        assert emb % 8 == 0, "emb must be a multiple of 8"
        return Dims(pca_vector=emb, embed_setting=(emb // 2, emb // 4, emb // 8, emb // 8),
                    category_label_num=category_label_num)


# BASELINE.json configs -> (B, H, T, emb)   (SURVEY §8d "Config -> shapes")
WORKLOADS = {
    "ref-default": dict(B=256, H=200, T=15, emb=64),
    "ref-test80": dict(B=80, H=200, T=15, emb=64),      # reference inference: model_test batches of 80 (train.py:107, verify.py:22)
    "C1-demo": dict(B=256, H=10, T=20, emb=256),
    "C2-small": dict(B=512, H=32, T=30, emb=256),
    "C3-large": dict(B=1024, H=50, T=30, emb=400),
    "C5-long": dict(B=256, H=128, T=64, emb=768),
}
