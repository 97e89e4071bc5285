"""json helpers used by the scale-factor loader (counterpart of gemnet/model/utils.py:4-40)."""
import json


def _check(path):
    if not isinstance(path, str) or not path.endswith(".json"):
        raise UserWarning(f"Path {path} is not a json-path.")


def read_json(path):
    _check(path)
    with open(path, "r") as f:
        return json.load(f)


def write_json(path, data):
    _check(path)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(data, f, ensure_ascii=False, indent=4)


def update_json(path, data):
    content = read_json(path)
    content.update(data)
    write_json(path, content)


def read_value_json(path, key):
    return read_json(path).get(key)
