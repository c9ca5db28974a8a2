"""reference: param/data_index.py:1-5."""


def get_data_index(data_name, params):
    for index, entry in enumerate(params):
        if entry['name'] == data_name:
            return index
    return -1
