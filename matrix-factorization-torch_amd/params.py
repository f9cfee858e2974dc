"""Constants mirrored from ``xfmr_rec/params.py:1-29`` (names and values kept)."""
# data
TARGET_COL = "rating"
ITEM_IDX_COL = "movie_rn"
ITEM_ID_COL = "movie_id"
ITEM_TEXT_COL = "movie_text"
USER_IDX_COL = "user_rn"
USER_ID_COL = "user_id"
USER_TEXT_COL = "user_text"

# model
BATCH_SIZE = 2**5
PADDING_IDX = 0
METRIC = {"name": "val/RetrievalNormalizedDCG", "mode": "max"}
TOP_K = 20

# serving / export
ITEMS_TABLE_NAME = "movies"
MODEL_NAME = "xfmr_rec"
PROCESSORS_JSON = "processors.json"
USERS_TABLE_NAME = "users"
# save() layout (xfmr_rec/params.py:24-29 has transformer/ + processors.json + lance_db/; tables replace both stores)
TOWERS_PATH = "towers.safetensors"
INDEX_PATH = "item_index.safetensors"
